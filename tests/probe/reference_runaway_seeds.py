"""BUILD CONTAINER ONLY (imports /root/reference through tests/golden/make_fixtures.py's stand-ins): the reference's own
Training.run() -- dr_constant_icml, modeuler, n_iwae 200, learning rate 0.01, 15 epochs = 105 steps -- at seeds 6..17 (0..5 were
run the same way): does its objective stay finite?  Output committed as profiles/r05_reference_runaway_seeds.log; the same runs
through this package: tests/probe/ref_seed_compare.py."""
import sys, os, tempfile
sys.path.insert(0, "/root/repo/tests/golden")
import make_fixtures as M
import numpy as np
M.install_standins()
sys.path.insert(0, M.REF)
os.chdir(M.REF)
os.environ["INFERENCE_RESULTS_DIR"] = tempfile.mkdtemp()
M.patch_merge_observations()
for seed in range(6, 18):
    fx = M.run_training_trace("dr_constant_icml", "modeuler", 200, 15, seed)
    l = fx["step_losses"]
    with open("/tmp/ref_seeds2.log", "a") as f:
        f.write("REFSEED %d n=%d min %.5g last %.5g valid %s\n" % (seed, len(l), l.min(), l[-1], fx["valid_elbo"]))
