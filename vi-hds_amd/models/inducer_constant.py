"""Arabinose-inducer models (plugin classes of the reference's models/inducer_constant.py:83-151); kernel: struct
InducerConstant in csrc/vihds_models.hpp.

The reference classes raise at construction (`init_with_params` does not exist on OdeModel, inducer_constant.py:85,
:119), so there is no runnable reference as shipped: the equations are restated from Inducer_Constant_RHS (:11-80).
inducer_constant_precisions is pinned against the MODIFIED reference (those construction defects repaired in memory,
tests/golden/make_fixtures.py --patched: inducer_constant_precisions_tiny_modeuler.npz) and the oracle; inducer_constant
(constant precisions) shares its RHS code."""
from vihds.ode import OdeModel
from vihds.precisions import ConstantPrecisions, NeuralPrecisions


class Inducer_Constant(OdeModel):
    model_key = "inducer_constant"
    observe_kind = "inducer"

    def __init__(self, config):
        super(Inducer_Constant, self).__init__(config)
        self.precisions = ConstantPrecisions(["prec_x", "prec_rfp", "prec_yfp", "prec_cfp"])
        self.species = ["OD", "RFP", "YFP", "F530", "F480"]
        self.n_species = 5
        self.version = 1

    def condition_theta(self, theta, dev_1hot, writer, epoch):
        return theta


class Inducer_Constant_Precisions(Inducer_Constant):
    model_key = "inducer_constant_precisions"

    def __init__(self, config):
        super(Inducer_Constant_Precisions, self).__init__(config)
        self.precisions = NeuralPrecisions(self.n_species, config.params.n_hidden_decoder_precisions, 4)

    def neural_weights(self):
        return self.precisions.flat_weights()

    def problem_kwargs(self, config):
        return {"n_hidden_prec": max(int(config.params.n_hidden_decoder_precisions), 0)}

    def summaries(self, writer, epoch):
        self.precisions.summaries(writer, epoch)
