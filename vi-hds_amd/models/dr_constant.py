"""Double-receiver white-box models (plugin classes of the reference's models/dr_constant.py:115-215).
The right-hand side, its adjoint and the theta -> effective-parameter map live in csrc/vihds_models.hpp
(struct DrConstant<VERSION>); these classes only describe the plugin surface."""
import torch

from vihds.ode import OdeModel
from vihds.precisions import ConstantPrecisions, NeuralPrecisions
from vihds.utils import variable_summaries

PREC = ["prec_x", "prec_rfp", "prec_yfp", "prec_cfp"]


class DR_Constant(OdeModel):
    model_key = "dr_constant"
    extra_theta_names = ("aR", "aS")

    def __init__(self, config):
        super(DR_Constant, self).__init__(config)
        self.precisions = ConstantPrecisions(PREC)
        self.species = ["OD", "RFP", "YFP", "CFP", "F530", "F480", "LuxR", "LasR"]
        self.n_species = 8
        self.version = 1
        self.aR = self.aS = None

    def condition_theta(self, theta, dev_1hot, writer, epoch):
        """aR, aS = device_conditioner(ones) (reference dr_constant.py:124-131)."""
        return self.condition_ones(theta, ["aR", "aS"], dev_1hot)

    def simulate(self, config, times, theta, conditions, dev_1hot, condition_on_device=True, observations=None):
        self.aR, self.aS = theta.aR, theta.aS
        return super(DR_Constant, self).simulate(config, times, theta, conditions, dev_1hot, condition_on_device,
                                                 observations)

    def summaries(self, writer, epoch):
        variable_summaries(writer, epoch, self.aR, "aR.conditioned")
        variable_summaries(writer, epoch, self.aS, "aS.conditioned")


class DR_Constant_V2(DR_Constant):
    model_key = "dr_constant_v2"

    def __init__(self, config):
        super(DR_Constant_V2, self).__init__(config)
        self.version = 2


class DR_Constant_Precisions(DR_Constant):
    model_key = "dr_constant_precisions"

    def __init__(self, config):
        super(DR_Constant_Precisions, self).__init__(config)
        self.precisions = NeuralPrecisions(self.n_species, config.params.n_hidden_decoder_precisions, 4)

    def neural_weights(self):
        return self.precisions.flat_weights()

    def problem_kwargs(self, config):
        return {"n_hidden_prec": max(int(config.params.n_hidden_decoder_precisions), 0)}

    def summaries(self, writer, epoch):
        super(DR_Constant_Precisions, self).summaries(writer, epoch)
        self.precisions.summaries(writer, epoch)


class DR_Constant_Precisions_V2(DR_Constant_Precisions):
    model_key = "dr_constant_precisions_v2"

    def __init__(self, config):
        super(DR_Constant_Precisions_V2, self).__init__(config)
        self.version = 2
