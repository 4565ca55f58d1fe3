"""Decoder (counterpart of the reference's vihds/decoders.py:13-45): model lookup by string, then
condition -> simulate -> expand_precisions -> observe.  The three middle calls are one kernel launch."""
from torch import nn

import models


class DecoderResult(tuple):
    """(x_states, x_predict, precisions) exactly as the reference returns them, plus the fused kernel's
    per-species log-likelihood so Training.cost does not re-read the [B,S,4,T] tensors."""

    log_p_by_species = None
    solution = None


class LazyDecoderResult(DecoderResult):
    """Training fast path: only the log-likelihood exists.  Unpacking or indexing the tuple computes
    (x_states, x_predict, precisions) the ordinary way first."""

    def __new__(cls, build):
        self = super(LazyDecoderResult, cls).__new__(cls, ())
        self._build, self._items = build, None
        return self

    def _get(self):
        if self._items is None:
            self._items = self._build()
        return self._items

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, k):
        return self._get()[k]

    def __len__(self):
        return 3


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
        fused = self.ode_model.solve_for_training(self.config, data.times, theta_conditioned, data.inputs,
                                                  data.dev_1hot, data.get("observations", None))
        if fused is not None:  # log-likelihood + unit-weight adjoint in one launch; the rest only on demand

            def build():
                full = fused.full()
                xs, prec = self.ode_model.expand_precisions(theta_conditioned, data.times, full.sol)
                return xs, self.ode_model.observe(full.sol, theta_conditioned), prec

            result = LazyDecoderResult(build)
            result.solution = fused
            result.log_p_by_species = fused.log_p_by_species
            return result, theta_conditioned
        solution = self.ode_model.simulate(
            self.config, data.times, theta_conditioned, data.inputs, data.dev_1hot,
            condition_on_device=self.condition_on_device, observations=data.get("observations", None),
        )
        last = self.ode_model._last
        if writer is not None:
            self.ode_model.summaries(writer, epoch)
        if not getattr(last, "has_x_predict", True):  # params.lazy_x_predict under no_grad: the tuple on demand

            def build_eval():
                xs, prec = self.ode_model.expand_precisions(theta_conditioned, data.times, solution)
                return xs, self.ode_model.observe(solution, theta_conditioned), prec

            result = LazyDecoderResult(build_eval)
        else:
            x_states, precisions = self.ode_model.expand_precisions(theta_conditioned, data.times, solution)
            x_predict = self.ode_model.observe(solution, theta_conditioned)
            result = DecoderResult((x_states, x_predict, precisions))
        result.solution = last
        result.log_p_by_species = last.log_p_by_species if getattr(last, "has_logp", False) else None
        return result, theta_conditioned
