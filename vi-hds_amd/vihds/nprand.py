"""numpy's legacy standard-normal stream from native code (csrc/host/vihds_nprand.cpp, libvihds_host.so): the SAME numbers
`np.random.randn(...).astype(np.float32)` returns -- and the same global RandomState afterwards -- several times faster
(MT19937 regenerated in vectorised runs, the polar method's attempts evaluated by a small thread pool).  The reference draws
u ~ N(0,1)[B,S,P] on the host every step (vihds/vae.py:22-24); with `u_rng: numpy` (the default: the reference's stream)
that draw was 25x the GPU's work for the step.

This is host-side plumbing, not the hot path: when the library is missing the call falls back to numpy itself."""
import atexit
import ctypes
import os

import numpy as np

_LIB = None
# (12 where the host has them: 0.115 ms for 252 000 normals on the GPU box against 0.138 at 8 and 0.12-0.15 at 16 -- the box allows
# 16 CPUs' worth of time, and the main thread and the helper want theirs; profiles/r04_nprand_timing.log)
_THREADS = int(os.environ.get("VIHDS_NPRAND_THREADS", str(12 if (os.cpu_count() or 1) >= 16 else min(8, os.cpu_count() or 1))))


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib", "libvihds_host.so")
        try:
            lib = ctypes.CDLL(path)
            lib.vihds_np_randn_f32.restype = ctypes.c_int
            lib.vihds_np_randn_f32.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                               ctypes.POINTER(ctypes.c_double), ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
            lib.vihds_np_randn_f32_start.restype = ctypes.c_int
            lib.vihds_np_randn_f32_start.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
                                                     ctypes.c_int]
            lib.vihds_np_randn_f32_wait.restype = ctypes.c_int
            lib.vihds_np_randn_f32_wait.argtypes = []
            # (a CPU without the vector extensions the library was compiled for, or numpy internals that moved: numpy itself)
            _LIB = lib if (_ADDR is not None and lib.vihds_host_cpu_ok()) else False
        except (OSError, AttributeError):
            _LIB = False
    return _LIB


def available():
    return bool(_lib())


# numpy's global MT19937 state, in place: key[624] then pos (numpy/random/src/mt19937/mt19937.h), exposed through the
# bit generator's ctypes interface.  get_state() / set_state() copy it through Python objects (0.1 ms per draw); the fast
# path below lets the native code read and advance it where it lives -- allowed only when nobody else has drawn since our
# last call (the state's fingerprint is unchanged) and that call left no cached second deviate (the legacy gauss cache is
# not reachable this way), which is every draw of an even number of normals in a row: the training loop's.
try:
    _BG = np.random.mtrand._rand._bit_generator
    _ADDR = _BG.ctypes.state_address
    _KEY = (ctypes.c_uint32 * 624).from_address(_ADDR)
    _POS = ctypes.c_int.from_address(_ADDR + 624 * 4)
except Exception:  # noqa: BLE001 -- private numpy internals: if they ever move, the native path is off, numpy draws
    _BG = _ADDR = _KEY = _POS = None
_LEFT = None


def _fingerprint():
    return (_POS.value, _KEY[0], _KEY[1], _KEY[396], _KEY[623])


_IN_FLIGHT = 0  # Draw.start()s whose finish() has not run yet (the library queues up to two)
_GEN = 0        # bumped whenever started draws were waited for by somebody other than their owner (_collect_stray)


def generation():
    """Token of the started draws' ownership: a draw started under generation g whose owner finds another generation later was
    waited for by _collect_stray in between -- its numbers are in place, and its finish() must not wait again."""
    return _GEN


def _collect_stray():
    """A started draw that nobody finished (its consumer never came: the stream has advanced by that unused draw, as
    hostdraws documents): wait for it, so that numpy's state is at rest before anybody reads or writes it."""
    global _IN_FLIGHT, _LEFT, _GEN
    if _IN_FLIGHT > 0:
        while _IN_FLIGHT > 0:
            _IN_FLIGHT -= 1
            _lib().vihds_np_randn_f32_wait()
        _LEFT = _fingerprint()
        _GEN += 1  # (whoever started them -- a graph's prefetch bookkeeping, vihds/hostdraws.py -- sees that they are done)


atexit.register(lambda: _collect_stray() if _IN_FLIGHT else None)  # (numpy's state must outlive a draw that is still running)


def randn_f32(shape, out=None):
    """float32 array of `shape` drawn from numpy's GLOBAL RandomState exactly as np.random.randn(*shape).astype(np.float32)
    draws it; `out`: a writable C-contiguous float32 buffer of that many elements (e.g. the numpy view of a pinned tensor)."""
    shape = tuple(int(v) for v in shape)
    n = int(np.prod(shape)) if shape else 1
    lib = _lib()
    _collect_stray()
    if out is None:
        out = np.empty(n, np.float32)
    flat = out.reshape(-1)
    if flat.dtype != np.float32 or flat.size != n or not flat.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous float32 buffer of %d elements" % n)
    if not lib or n == 0:
        flat[:] = np.random.randn(n).astype(np.float32) if n else 0
        return out.reshape(shape)
    global _LEFT
    with _BG.lock:
        if _LEFT is not None and n % 2 == 0 and _fingerprint() == _LEFT:
            chg, cg = ctypes.c_int(0), ctypes.c_double(0.0)
            rc = lib.vihds_np_randn_f32(_ADDR, ctypes.cast(_ADDR + 624 * 4, ctypes.POINTER(ctypes.c_int)), ctypes.byref(chg),
                                        ctypes.byref(cg), flat.ctypes.data, n, _THREADS)
            if rc != 0:
                raise RuntimeError("vihds_np_randn_f32 failed (%d)" % rc)
            _LEFT = _fingerprint()
            return out.reshape(shape)
    name, key, pos, has_gauss, gauss = np.random.get_state()
    key = np.ascontiguousarray(key, dtype=np.uint32).copy()
    cpos, chg, cg = ctypes.c_int(int(pos)), ctypes.c_int(int(has_gauss)), ctypes.c_double(float(gauss))
    rc = lib.vihds_np_randn_f32(key.ctypes.data, ctypes.byref(cpos), ctypes.byref(chg), ctypes.byref(cg), flat.ctypes.data, n,
                                _THREADS)
    if rc != 0:
        raise RuntimeError("vihds_np_randn_f32 failed (%d)" % rc)
    np.random.set_state((name, key, cpos.value, chg.value, cg.value))
    _LEFT = _fingerprint() if chg.value == 0 else None
    return out.reshape(shape)


class Draw(object):
    """A draw of `shape` normals from numpy's global stream into a host buffer: called, it draws now (`randn_f32`); `start`
    / `finish` run the same draw on the library's native helper thread -- beside the Python that queues the current step, which
    a Python helper thread cannot do (it needs the interpreter lock to start).  Between `start` and `finish` numpy's global
    generator belongs to the draw: nobody else may use it (vihds/hostdraws.py starts a draw only when its consumer is
    known to be the next user of the stream)."""

    def __init__(self, shape):
        self.shape = tuple(int(v) for v in shape)
        self.n = int(np.prod(self.shape)) if self.shape else 1

    def __call__(self, host):
        randn_f32(self.shape, out=host)

    def start(self, host):
        """True: the draw is running natively (call `finish` before reading `host`).  False: not started (no library, an odd
        count, or numpy's state is not the one our last draw left in place) -- the caller draws synchronously instead."""
        global _LEFT, _IN_FLIGHT
        lib = _lib()
        flat = host.reshape(-1)
        if not lib or self.n == 0 or self.n % 2 or flat.dtype != np.float32 or flat.size != self.n or not flat.flags.c_contiguous:
            return False
        # (with a draw of ours in flight the generator is ours already and its state in motion: nothing to compare; else it
        # must be the state our last draw left in place)
        if not _IN_FLIGHT and (_LEFT is None or _fingerprint() != _LEFT):
            return False
        rc = lib.vihds_np_randn_f32_start(_ADDR, _ADDR + 624 * 4, flat.ctypes.data, self.n, _THREADS)
        if rc != 0:  # (-2: two draws are queued already)
            return False
        _LEFT = None  # (the state is in motion until the last finish())
        _IN_FLIGHT += 1
        return True

    def generation(self):
        return _GEN

    def finish(self):
        """Wait for the OLDEST started draw."""
        global _LEFT, _IN_FLIGHT
        if _IN_FLIGHT <= 0:  # (collected already -- _collect_stray waited for it: the numbers are in place)
            return
        _IN_FLIGHT -= 1
        rc = _lib().vihds_np_randn_f32_wait()
        if rc != 0:
            raise RuntimeError("vihds_np_randn_f32_wait failed (%d)" % rc)
        if not _IN_FLIGHT:
            _LEFT = _fingerprint()
