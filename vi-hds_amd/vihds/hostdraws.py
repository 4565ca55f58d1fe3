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
    RING = 3

    def __init__(self):
        self.slots = []  # [device buffer, fill(host numpy view), offset in the arena, floats, prefetchable, started ahead]
        self.arena, self.used = None, 0
        self.noted = 0   # floats one step asked for (measuring mode: warm-up steps)
        self.pinned, self.views, self.events = None, None, [None] * self.RING
        self.pos, self.staged = 0, None  # next ring entry; the entry prefetch() has already started filling

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
        self.slots.append([buf, fill, self.used, n, prefetchable, False])
        self.used += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        return buf

    def _stage(self):
        """The next pinned copy of the arena, free to be written (the upload that last read it has run).  (Made at the first
        use: page-locked allocations are not permitted while a stream is capturing.)"""
        if self.pinned is None:
            self.pinned = [torch.empty(self.used, dtype=torch.float32).pin_memory() for _ in range(self.RING)]
            self.views = [[p[off: off + n].numpy() for (_b, _f, off, n, _p, _s) in self.slots] for p in self.pinned]
        k = self.pos
        self.pos = (k + 1) % self.RING
        if self.events[k] is not None:
            self.events[k].synchronize()
        return k

    def prefetch(self):
        """Start the prefetchable slots' NEXT draws on the native helper thread (fill.start / fill.finish: vihds/nprand.py).
        Only the caller knows that the next consumer of that random stream is this graph's next replay (Training.run: the
        next batch of the epoch); a prefetched draw that is never replayed has advanced the stream by one unused draw."""
        if self.staged is not None:
            return
        k, any_started = None, False
        for j, slot in enumerate(self.slots):
            if slot[4] and hasattr(slot[1], "start"):
                if k is None:
                    k = self._stage()
                slot[5] = bool(slot[1].start(self.views[k][j]))
                any_started = any_started or slot[5]
        if any_started:
            self.staged = k
        elif k is not None:
            self.pos = k  # (nothing could be started now: the refresh takes this entry itself)

    def refresh(self):
        k, self.staged = (self.staged if self.staged is not None else self._stage()), None
        for j, slot in enumerate(self.slots):
            if slot[5]:  # drawn ahead by prefetch()
                slot[5] = False
                slot[1].finish()
            else:
                slot[1](self.views[k][j])
        self.arena[: self.used].copy_(self.pinned[k], non_blocking=True)
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


def replay(graph, next_graph=None):
    """graph.replay() behind the refresh of the host draws it was captured with (Training attaches them as
    graph.host_draws).  next_graph: the graph whose replay is KNOWN to be the next consumer of the host streams (often the
    same one): its prefetchable draws start on the helper thread as soon as this replay is queued."""
    draws = getattr(graph, "host_draws", None)
    if draws:
        draws.refresh()
    graph.replay()
    nxt = getattr(next_graph, "host_draws", None) if next_graph is not None else None
    if nxt:
        nxt.prefetch()
