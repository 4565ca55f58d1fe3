"""Degrader models: the equations of the reference's models/degrader_constant.py (kernel: struct
DegraderConstant).  As for relay, the reference classes raise at construction (degrader_constant.py:17);
degrader_constant_precisions is pinned against the MODIFIED reference (construction defects repaired in memory,
tests/golden/make_fixtures.py --patched; tests/test_config5_parity.py) and the oracle, degrader_constant shares its RHS code."""
from vihds.ode import OdeModel
from vihds.precisions import ConstantPrecisions, NeuralPrecisions
from vihds.utils import variable_summaries


class Degrader_Constant(OdeModel):
    model_key = "degrader_constant"

    def __init__(self, config):
        super(Degrader_Constant, self).__init__(config)
        self.precisions = ConstantPrecisions(["prec_x", "prec_rfp", "prec_yfp", "prec_cfp"])
        self.species = ["OD", "RFP", "YFP", "CFP", "F530", "F480", "LuxR", "LasR", "AiiA", "C6", "C12"]
        self.n_species = 11
        self.version = 1
        self.aR = self.aS = None

    def condition_theta(self, theta, dev_1hot, writer, epoch):
        return theta

    def simulate(self, config, times, theta, conditions, dev_1hot, condition_on_device=True, observations=None):
        self.aR, self.aS = theta.aR, theta.aS
        return super(Degrader_Constant, self).simulate(config, times, theta, conditions, dev_1hot,
                                                       condition_on_device, observations)

    def summaries(self, writer, epoch):
        variable_summaries(writer, epoch, self.aR, "aR.conditioned")
        variable_summaries(writer, epoch, self.aS, "aS.conditioned")


class Degrader_Constant_Precisions(Degrader_Constant):
    model_key = "degrader_constant_precisions"

    def __init__(self, config):
        super(Degrader_Constant_Precisions, self).__init__(config)
        self.precisions = NeuralPrecisions(self.n_species, config.params.n_hidden_decoder_precisions, 4)

    def neural_weights(self):
        return self.precisions.flat_weights()

    def problem_kwargs(self, config):
        return {"n_hidden_prec": max(int(config.params.n_hidden_decoder_precisions), 0)}

    def summaries(self, writer, epoch):
        super(Degrader_Constant_Precisions, self).summaries(writer, epoch)
        self.precisions.summaries(writer, epoch)
