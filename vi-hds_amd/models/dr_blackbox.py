"""Black-box double-receiver model (plugin class of the reference's models/dr_blackbox.py:61-125): the RHS is
two small MLPs (NeuralStates, NeuralPrecisions) whose weights live here; the arithmetic runs in the kernel."""
import torch
import torch.nn as nn

from vihds import ops
from vihds.ode import OdeModel
from vihds.precisions import NeuralPrecisions
from vihds.utils import default_get_value, variable_summaries


class NeuralStates(nn.Module):
    """Weights of dx = sigmoid(W_p h) - sigmoid(W_d h) * x, h = relu(W_h [x, const]) (reference ode.py:119-146)."""

    def __init__(self, n_inputs, n_hidden, n_states, n_latents):
        super(NeuralStates, self).__init__()
        self.n_latents, self.n_states = n_latents, n_states
        self.states_hidden = nn.Linear(n_inputs, n_hidden)
        nn.init.xavier_uniform_(self.states_hidden.weight)
        self.states_production = nn.Linear(n_hidden, n_states)
        nn.init.xavier_uniform_(self.states_production.weight)
        self.states_degradation = nn.Linear(n_hidden, n_states)
        nn.init.xavier_uniform_(self.states_degradation.weight)

    def weight_tensors(self):
        mods = [self.states_hidden, self.states_production, self.states_degradation]
        return [t for m in mods for t in (m.weight, m.bias)]

    def flat_weights(self):
        return torch.cat([t.reshape(-1) for t in self.weight_tensors()])

    def summaries(self, writer, epoch):
        if writer is not None:
            for name in ["states_hidden", "states_production", "states_degradation"]:
                module = getattr(self, name)
                variable_summaries(writer, epoch, module.weight, name + "_weights", False)
                variable_summaries(writer, epoch, module.bias, name + "_bias", False)


class DR_Blackbox(OdeModel):
    model_key = "dr_blackbox"
    observe_kind = "direct"

    def __init__(self, config):
        super(DR_Blackbox, self).__init__(config)
        p = config.params
        self.n_x, self.n_y, self.n_z = p.n_x, p.n_y, p.n_z
        n_latents = self.n_x + self.n_y + self.n_z
        self.n_species = 4
        self.n_latent_species = p.n_latent_species
        self.n_hidden_precisions = p.n_hidden_decoder_precisions
        self.n_states = self.n_species + self.n_latent_species
        n_inputs = self.n_states + n_latents + self.n_treatments + self.device_depth
        self.precisions = NeuralPrecisions(n_inputs, self.n_hidden_precisions, 4, hidden_activation=nn.ReLU)
        self.species = ["OD", "RFP", "YFP", "CFP"]
        self.n_hidden = p.n_hidden_decoder
        self.init_latent_species = default_get_value(p, "init_latent_species", 0.001)
        self.init_prec = default_get_value(p, "init_prec", 0.00001)
        self.offset_layer = nn.Linear(self.device_depth, self.n_y)
        self.neural_states = NeuralStates(n_inputs, p.n_hidden_decoder, self.n_states, n_latents)
        # rows reserved behind the sampled parameters for the device-conditioned y (see condition_theta)
        self.extra_theta_names = tuple("y%d|device" % (i + 1) for i in range(self.n_y))
        self._flat = ops.FlatParameters()

    def condition_theta(self, theta, dev_1hot, writer, epoch):
        """y_i += offset_layer(dev_1hot)_i (reference dr_blackbox.py:86-96); re-binds the attribute only (theta.samples
        keeps the sampled y, which is what log q / log p are taken of).  When theta sits in a packed buffer with
        reserved rows, y + offset is written there in one launch and the simulator reads those rows; the gradient
        to y and to the offset layer is routed by ops.OdeSolveObserve (row_offset)."""
        if self.n_y == 0:
            return theta
        names = ["y%d" % (i + 1) for i in range(self.n_y)]
        packed = getattr(theta, "_packed", None)
        if packed is not None and packed.is_cuda and packed.is_contiguous() and theta.n_reserved_rows() >= self.n_y:
            rows = [theta._row_of.get(n) for n in names]
            base = len(theta.samples)
            if (None not in rows and rows == list(range(rows[0], rows[0] + self.n_y)) and rows[-1] < base
                    and not any(n in theta._rebound for n in names)):
                # one launch: offset layer + the add into the reserved rows (ops.OffsetRows); its token carries the
                # layer's gradients back from the simulator's backward (one more launch there)
                token = ops.OffsetRows.apply(self.offset_layer.weight, self.offset_layer.bias, dev_1hot, packed.detach(),
                                             rows[0], base)
                for i, n in enumerate(names):
                    theta.bind_reserved_row(n, base + i)
                object.__setattr__(theta, "_row_offset", (token, (rows[0], base, self.n_y, "linear")))
                return theta
        offset = self.offset_layer(dev_1hot)  # [B, n_y]
        for i, pname in enumerate(names):
            setattr(theta, pname, getattr(theta, pname) + offset[:, i: i + 1])
        return theta

    def neural_weights(self):
        return self._flat(self.neural_states.weight_tensors() + self.precisions.weight_tensors())

    def flat_weight_tensors(self):
        return self.neural_states.weight_tensors() + self.precisions.weight_tensors()

    def kernel_slots(self):
        """reference dr_blackbox.py:34-52: latents z (locals), x (globals), then the device-conditioned y."""
        return (["z%d" % (i + 1) for i in range(self.n_z)] + ["x%d" % (i + 1) for i in range(self.n_x)]
                + ["y%d" % (i + 1) for i in range(self.n_y)] + ["init_x", "init_rfp", "init_yfp", "init_cfp"])

    def problem_kwargs(self, config):
        return {
            "slots": self.kernel_slots(),
            "n_hidden_prec": int(self.n_hidden_precisions), "n_hidden_states": int(self.n_hidden),
            "n_latent_states": int(self.n_latent_species),
            "n_const": self.n_x + self.n_y + self.n_z + self.n_treatments + self.device_depth,
            "init_latent": float(self.init_latent_species), "init_prec": float(self.init_prec),
        }

    def summaries(self, writer, epoch):
        self.neural_states.summaries(writer, epoch)
        self.precisions.summaries(writer, epoch)
