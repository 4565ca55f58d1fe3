"""Debug model (plugin class of the reference's models/debug.py:11-53); kernel: struct DebugConstant in
csrc/vihds_models.hpp.  The reference class is stale (gen_reaction_equations keeps the pre-refactoring signature,
debug.py:35; observe indexes the time axis, :25-31) and cannot run: parity unpinned, observe taken as the evident
[OD, OD*s1, OD*s2, OD*s3]."""
from vihds.ode import OdeModel
from vihds.precisions import ConstantPrecisions


class Debug_Constant(OdeModel):
    model_key = "debug_constant"
    observe_kind = "direct"

    def __init__(self, config):
        super(Debug_Constant, self).__init__(config)
        self.precisions = ConstantPrecisions(["prec_x", "prec_rfp", "prec_yfp", "prec_cfp"])
        self.species = ["OD", "RFP", "YFP", "CFP"]
        self.n_species = len(self.species)

    def condition_theta(self, theta, dev_1hot, writer, epoch):
        return theta
