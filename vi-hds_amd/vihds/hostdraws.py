"""Host-side random draws inside captured steps.

The reference draws its normals on the host: u ~ N(0,1)[B,S,P] with numpy's global RandomState (vihds/vae.py:22-24) and the
device conditioner's weights with torch's CPU generator (a fresh DeviceConditioner per call, vihds/ode.py:48).  A hipGraph
replays device work only, so a captured step that keeps those streams (`u_rng: numpy`, `conditioner_rng: cpu` -- the
defaults) reads the numbers from STATIC device buffers: while a step is captured, every host draw registers a slot here
(what to draw, where it goes) instead of drawing; before every replay the slots are refreshed in registration order -- the
same draws, in the same order, from the same generators as the eager step -- through pinned staging buffers (used in turn,
each reused only after the copy that read it has run) and asynchronous copies that the replay queues behind."""
import torch

ACTIVE = None  # the HostDraws being recorded (set by Training while it captures)


class HostDraws(object):
    """The static buffers live in ONE arena that is allocated before the capture begins (`reserve`), sized from the warm-up
    steps that every capture runs first (`note` counts what a step draws): a buffer allocated from the graph's own memory
    pool while it captures was handed out again later in the same capture (measured: the conditioner's 56-byte weight
    buffer came back holding the evaluation's variance summaries).  The pinned side mirrors the arena -- three page-locked
    copies of it, used in turn: a refresh fills every slot's slice of one of them and uploads the whole with ONE copy."""

    ALIGN = 64  # floats
    RING = 5    # pinned copies: one being uploaded, one being filled now, up to AHEAD started ahead, one spare
    AHEAD = 2   # draws the native helper may have queued (vihds_np_randn_f32_start takes two)

    def __init__(self):
        self.slots = []  # [device buffer, fill(host numpy view), offset in the arena, floats, prefetchable]
        self.arena, self.used = None, 0
        self.noted = 0   # floats one step asked for (measuring mode: warm-up steps)
        self.pinned, self.views, self.events = None, None, [None] * self.RING
        self.pos = 0       # next ring entry
        self.staged = []   # (ring entry, the draw's ownership token) of prefetchable draws started ahead, oldest first

    def note(self, shape):
        n = 1
        for v in shape:
            n *= int(v)
        self.noted += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN

    def reserve(self, floats, device):
        self.arena = torch.empty(max(int(floats), self.ALIGN), device=device, dtype=torch.float32)
        self.used = 0

    def add(self, shape, device, fill, prefetchable=False):
        """Register a draw of `shape` float32 numbers; returns the static device buffer the captured kernels read.
        prefetchable: the draw may run on the helper thread ahead of its replay (numpy's stream, drawn by native code)."""
        n = 1
        for v in shape:
            n *= int(v)
        if self.arena is None or self.used + n > self.arena.numel():
            raise RuntimeError("host draws: the capture asks for more staged random numbers than its warm-up steps did "
                               "(%d floats reserved, %d in use, %d more wanted)"
                               % (0 if self.arena is None else self.arena.numel(), self.used, n))
        buf = self.arena[self.used: self.used + n].view(shape)
        self.slots.append([buf, fill, self.used, n, prefetchable and hasattr(fill, "start")])
        self.used += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        return buf

    def _stage(self):
        """The next pinned copy of the arena, free to be written (the upload that last read it has run).  (Made at the first
        use: page-locked allocations are not permitted while a stream is capturing.)"""
        if self.pinned is None:
            self.pinned = [torch.empty(self.used, dtype=torch.float32).pin_memory() for _ in range(self.RING)]
            self.views = [[p[off: off + n].numpy() for (_b, _f, off, n, _p) in self.slots] for p in self.pinned]
            self.upload = self.arena[: self.used]
        k = self.pos
        self.pos = (k + 1) % self.RING
        if self.events[k] is not None:
            self.events[k].synchronize()
        return k

    def prefetch(self, ahead=1):
        """Start the prefetchable slots' draws for the next `ahead` replays (at most AHEAD) on the native helper thread
        (fill.start / fill.finish: vihds/nprand.py), which goes from one straight into the next.  Only the caller knows that
        the next consumers of that random stream are this graph's next replays (Training.run: the epoch's next batches); a
        prefetched draw that is never replayed has advanced the stream by one unused draw."""
        n_pre = sum(1 for slot in self.slots if slot[4])
        if n_pre != 1:  # (none to start -- or several streams to interleave, which the one queue cannot order)
            return
        j = next(i for i, slot in enumerate(self.slots) if slot[4])
        while len(self.staged) < min(int(ahead), self.AHEAD):
            k = self._stage()
            fill = self.slots[j][1]
            if not fill.start(self.views[k][j]):
                self.pos = k  # (not startable now: the refresh takes this entry itself)
                break
            self.staged.append((k, fill.generation() if hasattr(fill, "generation") else None))

    def refresh(self):
        ahead = bool(self.staged)
        k, token = self.staged.pop(0) if ahead else (self._stage(), None)
        for j, slot in enumerate(self.slots):
            if ahead and slot[4]:  # drawn ahead by prefetch()
                # Another consumer of the same host stream (an evaluate() refresh, another graph's refresh, an eager draw, a
                # training-state snapshot) may have had to wait for this draw in the meantime (nprand._collect_stray): the
                # numbers are in the staging buffer then and there is nothing left to wait for -- waiting again would take
                # somebody else's draw off the helper's queue (the two books could desynchronise)
                if not hasattr(slot[1], "generation") or slot[1].generation() == token:
                    slot[1].finish()
            else:
                slot[1](self.views[k][j])
        self.upload.copy_(self.pinned[k], non_blocking=True)
        if self.events[k] is None:
            self.events[k] = torch.cuda.Event()
        self.events[k].record()

    def __bool__(self):
        return bool(self.slots)


def capturing():
    return ACTIVE is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def note(shape):
    """A host draw of `shape` happens now, eagerly: if a capture is being prepared, count it."""
    if ACTIVE is not None:
        ACTIVE.note(shape)


def replay(graph, next_graph=None, ahead=1):
    """graph.replay() behind the refresh of the host draws it was captured with (Training attaches them as
    graph.host_draws).  next_graph: the graph whose replays are KNOWN to be the next `ahead` consumers of the host streams
    (often the same one): its prefetchable draws start on the helper thread as soon as this replay is queued."""
    draws = getattr(graph, "host_draws", None)
    if draws:
        draws.refresh()
    graph.replay()
    nxt = getattr(next_graph, "host_draws", None) if next_graph is not None else None
    if nxt and (nxt is draws or not (draws and draws.staged)):
        nxt.prefetch(ahead)
