"""PRPR models (plugin classes of the reference's models/prpr_constant.py:72-130); kernel: struct PrprConstant."""
from vihds.ode import OdeModel
from vihds.precisions import ConstantPrecisions, NeuralPrecisions


class PRPR_Constant(OdeModel):
    model_key = "prpr_constant"

    def __init__(self, config):
        super(PRPR_Constant, self).__init__(config)
        self.precisions = ConstantPrecisions(["prec_x", "prec_rfp", "prec_yfp", "prec_cfp"])
        self.species = ["OD", "RFP", "YFP", "CFP", "F530", "F480"]
        self.n_species = 6
        self.version = 1

    def condition_theta(self, theta, dev_1hot, writer, epoch):
        return theta


class PRPR_Constant_Precisions(PRPR_Constant):
    model_key = "prpr_constant_precisions"

    def __init__(self, config):
        super(PRPR_Constant_Precisions, self).__init__(config)
        self.precisions = NeuralPrecisions(self.n_species, config.params.n_hidden_decoder_precisions, 4)

    def neural_weights(self):
        return self.precisions.flat_weights()

    def problem_kwargs(self, config):
        return {"n_hidden_prec": max(int(config.params.n_hidden_decoder_precisions), 0)}

    def summaries(self, writer, epoch):
        self.precisions.summaries(writer, epoch)
