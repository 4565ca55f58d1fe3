"""Small host-side helpers shared by the vihds package (counterpart of the reference's vihds/utils.py)."""
import os

import numpy as np


class AttrDict(dict):
    """dict with attribute access -- what the reference gets from the third-party `munch` package
    (vihds/config.py:9); written here so the path has no dependency the image lacks."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def attrify(x):
    """Recursive dict -> AttrDict (the reference's munchify)."""
    if isinstance(x, dict):
        return AttrDict((k, attrify(v)) for k, v in x.items())
    if isinstance(x, (list, tuple)):
        return type(x)(attrify(v) for v in x)
    return x


munchify = attrify  # name used by callers of the reference API


def default_get_value(dct, key, default_value, verbose=False):
    """reference vihds/utils.py:42-47"""
    if key in dct:
        return dct[key]
    if verbose:
        print("%s using default %s" % (key, str(default_value)))
    return default_value


def variable_summaries(writer, epoch, var, name, plot_histograms=False):
    """TensorBoard scalar summaries of a tensor (reference vihds/utils.py:30-39)."""
    if writer is None:
        return
    mean = var.mean()
    writer.add_scalar(name + "/mean", mean, epoch)
    writer.add_scalar(name + "/stddev", (var - mean).pow(2).mean().sqrt(), epoch)
    writer.add_scalar(name + "/max", var.max(), epoch)
    writer.add_scalar(name + "/min", var.min(), epoch)
    if plot_histograms:
        writer.add_histogram(name + "/histogram", var, epoch)


class TrainingLogData:
    """Timers and ELBO history collected during training (reference vihds/utils.py:50-62)."""

    def __init__(self):
        self.training_elbo_list = []
        self.validation_elbo_list = []
        self.batch_feed_time = 0.0
        self.batch_train_time = 0.0
        self.total_train_time = 0.0
        self.total_test_time = 0.0
        self.n_test = 0
        self.max_val_elbo = -float("inf")


_PINNED = {}


def _pinned(n, dtype):
    """A reusable pinned staging buffer of at least n elements (pageable device->host copies run at a third of the rate)."""
    import torch

    key = str(dtype)
    buf = _PINNED.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1), dtype=dtype, pin_memory=True)
        _PINNED[key] = buf
    return buf[:n]


def _to_host(tensors, stack=False):
    """numpy copies of a list of tensors through ONE device->host transfer (pinned staging).  stack: the tensors have one
    shape and the result is a single [len, *shape] array (rows of one device buffer are copied as one block, not gathered)."""
    import torch

    tensors = [t.detach() for t in tensors]
    if not tensors:
        return np.zeros((0,)) if stack else []
    if not tensors[0].is_cuda or len({t.dtype for t in tensors}) != 1:
        out = [t.cpu().numpy() for t in tensors]
        return np.array(out) if stack else out
    t0 = tensors[0]
    step = t0.numel() * t0.element_size()
    contiguous_rows = (stack and all(t.shape == t0.shape and t.is_contiguous() for t in tensors)
                       and all(t.data_ptr() == t0.data_ptr() + k * step for k, t in enumerate(tensors)))
    if contiguous_rows:  # e.g. theta: consecutive rows of the packed [R,B,S] buffer
        src = torch.as_strided(t0, (len(tensors) * t0.numel(),), (1,))
    else:
        src = torch.cat([t.reshape(-1) for t in tensors])
    host = _pinned(src.numel(), src.dtype)
    host.copy_(src, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    flat = host.numpy().copy()  # (the staging buffer is reused by the next call)
    if stack and all(t.shape == t0.shape for t in tensors):
        return flat.reshape((len(tensors),) + tuple(t0.shape))
    out, o = [], 0
    for t in tensors:
        n = t.numel()
        out.append(flat[o:o + n].reshape(tuple(t.shape)))
        o += n
    return np.array(out) if stack else out


class Results:
    """Evaluation results with the reference's attribute names and on-disk format (vihds/utils.py:65-156).

    The importance-weighted summaries are computed on the GPU by vihds_iw_summaries (no [B,S,.,T] host
    copies); only the [B,.,T] results travel to the host."""

    _ARRAYS = ["q_values", "theta", "elbo", "iw_predict_mu", "iw_predict_std", "iw_states", "iw_variance"]

    def __init__(self):
        self.species_names = None
        self.q_names = None
        self.q_values = None
        self._theta_host = None
        self._theta_dev = None
        self.elbo = None
        self.iw_predict_mu = None
        self.iw_predict_std = None
        self.iw_states = None
        self.iw_variance = None

    def init_from_device(self, species_names, q, theta, elbo, summaries):
        self.species_names = species_names
        self.q_names = q.get_tensor_names()
        # the reference's Results holds numpy arrays (utils.py:79-99): ONE device->host copy per group of tensors
        # instead of one (synchronising) copy per distribution parameter and per theta row
        self.q_values = np.array(_to_host(q.get_tensors()), dtype=object)
        # theta [P,B,S] is by far the largest member (33 MB at 234 rows x 1000 samples: the pass's device->host copy was
        # 80 % of its time) and only dump() / a reader of `.theta` ever looks at it: it stays on the device until then
        self._theta_dev, self._theta_host = list(theta.get_tensors()), None
        mu, sd, st, var = summaries
        (self.elbo, self.iw_predict_mu, self.iw_predict_std, self.iw_states,
         self.iw_variance) = _to_host([elbo, mu, sd, st, var])

    def init_from_staged(self, species_names, staged, host):
        """The same members from the outputs of a captured evaluation pass (Training.evaluate).  `host` is the pinned buffer
        the pass's `flat` -- the q tables, the ELBO and the four summaries back to back -- is on its way into; the numpy
        members are VIEWS of it (no second copy on the host) until Training.evaluate hands the buffer to a later pass
        (detach_host).  The theta rows stay in the graph's memory pool until somebody reads them or the next replay is about
        to overwrite them (detach_theta)."""
        import torch

        self.species_names = species_names
        self.q_names = staged["q_names"]
        self._theta_dev, self._theta_host = list(staged["theta_rows"]), None
        torch.cuda.current_stream().synchronize()
        arr = host.numpy()
        o, qv = 0, []
        for shp in staged["q_shapes"]:
            n = int(np.prod(shp)) if len(shp) else 1
            qv.append(arr[o:o + n].reshape(shp))
            o += n
        self.q_values = np.array(qv, dtype=object)
        # the ELBO is a scalar callers KEEP (log_data.max_val_elbo, the elbo lists of _evaluate_elbo_and_plot): it gets
        # storage of its own at once, never a view of a staging slot a later pass rewrites
        self.elbo = np.array(arr[o], dtype=arr.dtype)
        o += 1
        out = []
        for shp in staged["summary_shapes"]:
            n = int(np.prod(shp))
            out.append(arr[o:o + n].reshape(shp))
            o += n
        self.iw_predict_mu, self.iw_predict_std, self.iw_states, self.iw_variance = out
        self._host_views = True

    def detach_host(self):
        """Give the numpy members storage of their own (they were views of a staging buffer that is about to be reused)."""
        if not getattr(self, "_host_views", False):
            return
        self._host_views = False
        qv = [np.array(v, copy=True) for v in self.q_values]
        self.q_values = np.array(qv, dtype=object)
        self.iw_predict_mu, self.iw_predict_std = self.iw_predict_mu.copy(), self.iw_predict_std.copy()
        self.iw_states, self.iw_variance = self.iw_states.copy(), self.iw_variance.copy()

    def detach_theta(self):
        """Give the theta samples storage of their own (a device copy) if they are still views of somebody else's buffer
        that is about to be rewritten."""
        import torch

        rows = self._theta_dev
        if self._theta_host is not None or not rows:
            return
        t0 = rows[0]
        step = t0.numel() * t0.element_size()
        if all(t.shape == t0.shape and t.is_contiguous() and t.data_ptr() == t0.data_ptr() + k * step
               for k, t in enumerate(rows)):
            block = torch.as_strided(t0, (len(rows),) + tuple(t0.shape), (t0.numel(),) + tuple(t0.stride())).clone()
        else:
            block = torch.stack(rows)
        self._theta_dev = list(block.unbind(0))

    @property
    def theta(self):
        """numpy [P,B,S] as in the reference's Results (utils.py:83); copied from the device on first use."""
        if self._theta_host is None and self._theta_dev is not None:
            self._theta_host = _to_host(self._theta_dev, stack=True)
            self._theta_dev = None
        return self._theta_host

    @theta.setter
    def theta(self, value):
        self._theta_host, self._theta_dev = value, None

    def dump(self, location=".vihds_cache"):
        os.makedirs(location, exist_ok=True)
        for base, data in (("species_names", self.species_names), ("q_names", self.q_names)):
            np.savetxt(os.path.join(location, base + ".csv"), np.array(data, dtype=str), delimiter=",", fmt="%s")
        for base in self._ARRAYS:
            np.save(os.path.join(location, base + ".npy"), getattr(self, base))

    def load(self, location=".vihds_cache"):
        self.species_names = np.loadtxt(os.path.join(location, "species_names.csv"), dtype=str, delimiter=",")
        self.q_names = np.loadtxt(os.path.join(location, "q_names.csv"), dtype=str, delimiter=",")
        for base in self._ARRAYS:
            setattr(self, base, np.load(os.path.join(location, base + ".npy"), allow_pickle=True))
