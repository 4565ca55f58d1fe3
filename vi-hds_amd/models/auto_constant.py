"""Autofluorescence models (plugin classes of the reference's models/auto_constant.py:66-132); kernel: struct
AutoConstant in csrc/vihds_models.hpp."""
from vihds.ode import OdeModel
from vihds.precisions import ConstantPrecisions, NeuralPrecisions


class Auto_Constant(OdeModel):
    model_key = "auto_constant"
    observe_kind = "direct"

    def __init__(self, config):
        super(Auto_Constant, self).__init__(config)
        self.precisions = ConstantPrecisions(["prec_x", "prec_rfp", "prec_yfp", "prec_cfp"])
        self.species = ["OD", "RFP", "F530", "F480"]
        self.n_species = 4
        self.version = 1

    def condition_theta(self, theta, dev_1hot, writer, epoch):
        return theta


class Auto_Constant_Precisions(Auto_Constant):
    model_key = "auto_constant_precisions"

    def __init__(self, config):
        super(Auto_Constant_Precisions, self).__init__(config)
        self.precisions = NeuralPrecisions(self.n_species, config.params.n_hidden_decoder_precisions, 4)

    def neural_weights(self):
        return self.precisions.flat_weights()

    def problem_kwargs(self, config):
        return {"n_hidden_prec": max(int(config.params.n_hidden_decoder_precisions), 0)}

    def summaries(self, writer, epoch):
        self.precisions.summaries(writer, epoch)
