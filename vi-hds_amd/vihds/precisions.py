"""Observation precisions (counterpart of the reference's vihds/precisions.py)."""
import torch
from torch import nn

from vihds.utils import variable_summaries


class ConstantPrecisions(nn.Module):
    """Precisions are four theta entries, constant in time (reference precisions.py:19-38)."""

    def __init__(self, precision_vars):
        super(ConstantPrecisions, self).__init__()
        self.dynamic = False
        self.precision_vars = precision_vars

    def expand(self, theta, n_times, x_states):
        # [B,S,4,T]; a stride-0 view over T instead of the reference's materialising .repeat (same values), and
        # when the four precisions are consecutive rows of theta's packed buffer, a pure view of those rows
        rows = [getattr(theta, "_row_of", {}).get(v) for v in self.precision_vars]
        packed = getattr(theta, "_packed", None)
        if packed is not None and None not in rows and rows == list(range(rows[0], rows[0] + len(rows))) \
                and not any(v in theta._rebound for v in self.precision_vars):
            p = packed[rows[0]: rows[0] + len(rows)].permute(1, 2, 0)
        else:
            p = torch.stack([getattr(theta, v) for v in self.precision_vars], dim=-1)
        return x_states, p.unsqueeze(3).expand(-1, -1, -1, n_times)

    def summaries(self, _writer, _epoch):
        pass


class NeuralPrecisions(nn.Module):
    """d prec/dt = sigmoid(prod(.)) - sigmoid(degr(.)) * prec with the wiring of reference precisions.py:41-74.
    The weights live here (so optimisers and summaries see them); the arithmetic runs inside the ODE kernels."""

    def __init__(self, n_inputs, n_hidden_precisions, n_outputs, inverse=False, hidden_activation=nn.Tanh):
        super(NeuralPrecisions, self).__init__()
        print("- Initialising neural precisions with %d hidden layers" % n_hidden_precisions)
        if inverse:
            raise NotImplementedError("inverse neural precisions are not used by any reference model")
        self.dynamic = True
        self.inverse = inverse
        self._flat = None
        self.n_inputs, self.n_outputs, self.n_hidden = n_inputs, n_outputs, n_hidden_precisions
        self.activation = "relu" if hidden_activation is nn.ReLU else "tanh"
        n_in = n_inputs + 1
        if n_hidden_precisions < 1:
            self.prec_production = nn.Linear(n_in, n_outputs)
            nn.init.xavier_uniform_(self.prec_production.weight)
            self.prec_degradation = nn.Linear(n_in, n_outputs)
            nn.init.xavier_uniform_(self.prec_degradation.weight)
        else:
            self.prec_hidden = nn.Linear(n_in, n_hidden_precisions)
            nn.init.xavier_uniform_(self.prec_hidden.weight)
            self.prec_production = nn.Linear(n_hidden_precisions, n_outputs)
            nn.init.xavier_uniform_(self.prec_production.weight, gain=0.5)
            self.prec_degradation = nn.Linear(n_hidden_precisions, n_outputs)
            nn.init.xavier_uniform_(self.prec_degradation.weight, gain=1)

    def weight_tensors(self):
        """Parameters in the kernel's buffer order: [hid_w, hid_b,] prod_w, prod_b, degr_w, degr_b (row-major)."""
        mods = ([self.prec_hidden] if self.n_hidden >= 1 else []) + [self.prec_production, self.prec_degradation]
        return [t for m in mods for t in (m.weight, m.bias)]

    def flat_weights(self):
        """The kernel's weight buffer: the parameters themselves, kept back to back (ops.FlatParameters)."""
        if self._flat is None:
            from vihds import ops
            object.__setattr__(self, "_flat", ops.FlatParameters())
        return self._flat(self.weight_tensors())

    def expand(self, theta, _n_times, x_states):
        return x_states[:, :, : -self.n_outputs, :], x_states[:, :, -self.n_outputs:, :]

    def summaries(self, writer, epoch):
        if writer is not None:
            for name in ["prec_hidden", "prec_production", "prec_degradation"]:
                if hasattr(self, name):
                    module = getattr(self, name)
                    variable_summaries(writer, epoch, module.weight, name + "_weights", False)
                    variable_summaries(writer, epoch, module.bias, name + "_bias", False)
