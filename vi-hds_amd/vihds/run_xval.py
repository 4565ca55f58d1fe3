"""Command-line entry point with the reference's flags (vihds/run_xval.py:17-91):
    python vi-hds_amd/vihds/run_xval.py --experiment=X --gpu=0 specs/dr_constant_icml.yaml
"""
from __future__ import absolute_import

import argparse
import os
import sys

if __package__ in (None, ""):  # executed as a script: make `vihds` and `models` importable
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vihds.config import Config, Trainer  # noqa: E402
from vihds.datasets import build_datasets  # noqa: E402
from vihds.parameters import Parameters  # noqa: E402
from vihds.training import Training  # noqa: E402
from vihds.vae import build_model  # noqa: E402


def create_parser(with_split):
    """The reference's command line (run_xval.py:17-57): same flags, types and defaults -- that is the CLI boundary a spec /
    script written for vi-hds relies on; the help texts are this package's own."""
    parser = argparse.ArgumentParser(description="vi-hds on MI355X: IWAE training of one cross-validation split")
    parser.add_argument("yaml", type=str, help="experiment definition (a specs/*.yaml file of the reference loads unchanged)")
    parser.add_argument("--experiment", type=str, default="unnamed",
                        help="experiment name: sub-directory for TensorBoard logs and cached results")
    parser.add_argument("--seed", type=int, default=None, help="seed of numpy / torch (0 when omitted)")
    parser.add_argument("--epochs", type=int, default=1000, help="passes over the training rows")
    parser.add_argument("--test_epoch", type=int, default=20, help="evaluate train / validation ELBO every this many epochs")
    parser.add_argument("--plot_epoch", type=int, default=100, help="accepted for compatibility (figures are out of scope here)")
    parser.add_argument("--train_samples", type=int, default=200, help="IWAE samples per data row in a training step")
    parser.add_argument("--test_samples", type=int, default=1000, help="IWAE samples per data row in an evaluation pass")
    parser.add_argument("--dreg", type=bool, default=True, help="accepted for compatibility (the reference never reads it either)")
    parser.add_argument("--precision_hidden_layers", type=int, default=None,
                        help="hidden units of the neural-precision network (overrides n_hidden_decoder_precisions)")
    parser.add_argument("--verbose", action="store_true", default=False, help="print the parameter tables while they are built")
    parser.add_argument("--gpu", type=int, default=None,
                        help="index of the MI355X to run on; without it only host-side construction works (no CPU kernels)")
    if with_split:
        group = parser.add_mutually_exclusive_group()
        group.add_argument("--heldout", type=str, help="hold out every row of this device, e.g. R33S32_Y81C76")
        group.add_argument("--split", type=int, default=1, help="which of the `folds` cross-validation splits is validated on")
        group.add_argument("--figures", action="store_true", default=False, help="accepted for compatibility (no figures here)")
    parser.add_argument("--folds", type=int, default=4, help="number of cross-validation splits")
    return parser


def run_on_split(args, settings, split=None):
    """Run one train-test split (reference run_xval.py:60-72)."""
    if getattr(args, "heldout", None):
        print("Heldout device is %s" % args.heldout)
    else:
        args.heldout = None
        if split is not None:
            args.split = split
    data = build_datasets(args, settings)
    parameters = Parameters(settings.params)
    model = build_model(args, settings, data, parameters)
    training = Training(args, settings, data, parameters, model)
    return data, training.run()


def main():
    parser = create_parser(True)
    args = parser.parse_args()
    settings = Config(args)
    settings.trainer = Trainer(args, add_timestamp=True)
    data_pair, val_results = run_on_split(args, settings)
    if val_results is not None:
        # cross-validation merge / figures (vihds/xval.py, plotting.py) are post-processing outside this path
        print("validation iwae-elbo of best epoch: %s" % val_results.elbo)


if __name__ == "__main__":
    main()
