"""Relay models: the equations of the reference's models/relay_constant.py (kernel: struct RelayConstant).

The reference classes cannot be constructed at this commit (relay_constant.py:17 passes five arguments to
OdeFunc.__init__, :201 calls a non-existent init_with_params -- SURVEY.md 2.1).  Parity of relay_constant_precisions (the
one relay spec the reference ships) is against the MODIFIED reference -- exactly those two construction defects repaired
in memory, its own forward() untouched (tests/golden/make_fixtures.py --patched; tests/test_config5_parity.py: trajectories,
precision states, log-likelihood, loss, every theta and network-weight gradient) -- and against the oracle; relay_constant
(constant precisions, no spec in the reference) shares the same RHS code."""
from vihds.ode import OdeModel
from vihds.precisions import ConstantPrecisions, NeuralPrecisions
from vihds.utils import variable_summaries


class Relay_Constant(OdeModel):
    model_key = "relay_constant"

    def __init__(self, config):
        super(Relay_Constant, self).__init__(config)
        self.precisions = ConstantPrecisions(["prec_x", "prec_rfp", "prec_yfp", "prec_cfp"])
        self.species = ["OD", "RFP", "YFP", "CFP", "F530", "F480", "LuxR", "LasR", "LuxI", "LasI", "C6", "C12"]
        self.n_species = 12
        self.version = 1
        self.aR = self.aS = None

    def condition_theta(self, theta, dev_1hot, writer, epoch):
        # aR / aS are ordinary (global_conditioned) parameters of the relay spec (relay_constant.py:69-70)
        return theta

    def simulate(self, config, times, theta, conditions, dev_1hot, condition_on_device=True, observations=None):
        self.aR, self.aS = theta.aR, theta.aS
        return super(Relay_Constant, self).simulate(config, times, theta, conditions, dev_1hot, condition_on_device,
                                                    observations)

    def summaries(self, writer, epoch):
        variable_summaries(writer, epoch, self.aR, "aR.conditioned")
        variable_summaries(writer, epoch, self.aS, "aS.conditioned")


class Relay_Constant_Precisions(Relay_Constant):
    model_key = "relay_constant_precisions"

    def __init__(self, config):
        super(Relay_Constant_Precisions, self).__init__(config)
        self.precisions = NeuralPrecisions(self.n_species, config.params.n_hidden_decoder_precisions, 4)

    def neural_weights(self):
        return self.precisions.flat_weights()

    def problem_kwargs(self, config):
        return {"n_hidden_prec": max(int(config.params.n_hidden_decoder_precisions), 0)}

    def summaries(self, writer, epoch):
        super(Relay_Constant_Precisions, self).summaries(writer, epoch)
        self.precisions.summaries(writer, epoch)
