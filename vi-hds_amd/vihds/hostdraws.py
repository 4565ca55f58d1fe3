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
    buffer came back holding the evaluation's variance summaries)."""

    ALIGN = 64  # floats

    def __init__(self):
        self.slots = []  # (device buffer, fill(host numpy view), [pinned buffers], [events], position)
        self.arena, self.used = None, 0
        self.noted = 0   # floats one step asked for (measuring mode: warm-up steps)

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
        self.used += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        # (the pinned staging buffers are made at the first refresh: page-locked allocations are not permitted while a
        # stream is capturing)
        self.slots.append([buf, fill, None, [None, None, None], 0, prefetchable, None])
        return buf

    def _stage(self, slot):
        """The slot's next pinned buffer, free to be written (the copy that last read it has run)."""
        buf, _fill, pinned, events, k = slot[:5]
        if pinned is None:
            pinned = slot[2] = [torch.empty(buf.numel(), dtype=torch.float32).pin_memory() for _ in range(3)]
        slot[4] = (k + 1) % len(pinned)
        if events[k] is not None:
            events[k].synchronize()
        return k

    def prefetch(self):
        """Start the prefetchable slots' NEXT draws on the native helper thread (fill.start / fill.finish: vihds/nprand.py).
        Only the caller knows that the next consumer of that random stream is this graph's next replay (Training.run: the
        next batch of the epoch); a prefetched draw that is never replayed has advanced the stream by one unused draw."""
        for slot in self.slots:
            if slot[5] and slot[6] is None and hasattr(slot[1], "start"):
                k = self._stage(slot)
                if slot[1].start(slot[2][k].numpy()):
                    slot[6] = k
                else:  # not startable now: drawn at the refresh, into the buffer that was just staged
                    slot[4] = k

    def refresh(self):
        for slot in self.slots:
            buf, fill, events = slot[0], slot[1], slot[3]
            if slot[6] is not None:  # drawn ahead by prefetch()
                k, slot[6] = slot[6], None
                fill.finish()
            else:
                k = self._stage(slot)
                fill(slot[2][k].numpy())
            pinned = slot[2]
            buf.copy_(pinned[k].view(buf.shape), non_blocking=True)
            if events[k] is None:
                events[k] = torch.cuda.Event()
            events[k].record()

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
