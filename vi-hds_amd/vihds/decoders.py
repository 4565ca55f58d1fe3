"""Decoder (counterpart of the reference's vihds/decoders.py:13-45): model lookup by string, then
condition -> simulate -> expand_precisions -> observe.  The three middle calls are one kernel launch."""
from torch import nn

import models


class DecoderResult(tuple):
    """(x_states, x_predict, precisions) exactly as the reference returns them, plus the fused kernel's
    per-species log-likelihood so Training.cost does not re-read the [B,S,4,T] tensors."""

    log_p_by_species = None
    solution = None


class Decoder(nn.Module):
    def __init__(self, config, condition_on_device):
        super(Decoder, self).__init__()
        print("Initialising decoder")
        ode_model_class = models.LOOKUP[config.model]
        self.ode_model = ode_model_class(config)
        self.state_names = self.ode_model.species
        self.condition_on_device = condition_on_device
        self.config = config

    def forward(self, theta, data, writer, epoch):
        if self.condition_on_device:
            theta_conditioned = self.ode_model.condition_theta(theta, data.dev_1hot, writer, epoch)
        else:
            theta_conditioned = theta
        solution = self.ode_model.simulate(
            self.config, data.times, theta_conditioned, data.inputs, data.dev_1hot,
            condition_on_device=self.condition_on_device, observations=data.get("observations", None),
        )
        x_states, precisions = self.ode_model.expand_precisions(theta_conditioned, data.times, solution)
        x_predict = self.ode_model.observe(solution, theta_conditioned)
        if writer is not None:
            self.ode_model.summaries(writer, epoch)
        result = DecoderResult((x_states, x_predict, precisions))
        last = self.ode_model._last
        result.solution = last
        result.log_p_by_species = last.log_p_by_species if getattr(last, "has_logp", False) else None
        return result, theta_conditioned
