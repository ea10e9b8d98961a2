"""Handle / dataset caches shared by the reference-shaped front ends (funcs, mcmc_eqns, synth, LogProb)."""
import contextlib
import hashlib
import threading
import zlib

import numpy as np

from . import _capi

_lock = threading.Lock()
_handles = {}          # key -> Engine, in least-recently-used order (dicts keep insertion order)
MAX_ENGINES = 8        # each holds a stream, the grid and its workspace on the GPU: a keyword sweep must not pile them up


def grid(GRBtype=None):
    """Time grid selection of magnetar/funcs.py:132-141 (same ValueError text)."""
    if GRBtype is not None and GRBtype == "S":
        return np.logspace(-3.0, 6.0, num=10001, base=10.0)
    if GRBtype is None or GRBtype == "L":
        return np.logspace(0.0, 6.0, num=10001, base=10.0)
    raise ValueError("Please provide a valid value for GRBtype.\nOptions are: L, S, or None.")


def _as_f64(a):
    """`a` as a contiguous 1-D-compatible float64 ndarray, without a copy where it already is one (pandas: its values)."""
    v = getattr(a, "values", a)
    if type(v) is np.ndarray and v.dtype == np.float64 and v.flags.c_contiguous:
        return v
    return np.ascontiguousarray(a, dtype=np.float64)


def _cfg_key(cfg):
    return tuple(getattr(cfg, f[0]) for f in cfg._fields_)


class Engine:
    """One mp_handle plus a content-addressed cache of the datasets registered on it."""

    def __init__(self, cfg, tgrid, device=-1):
        self.handle = _capi.Handle(cfg, tgrid, device)
        self._slots = {}      # digest -> slot
        self._order = []      # LRU of digests
        self._prior_key = None
        self.lock = threading.RLock()   # re-entrant: a thread may call clear() (or evaluate again) inside its own use() block
        self.pins = 0         # callers between use()'s entry and exit (guarded by the module lock): never evicted

    def dataset_slot(self, x, y, yerr):
        """Slot of the light curve (x, y, yerr) on this handle, registering it on first sight.  Content-addressed: the same
        numbers give the same slot whatever object carries them, and an array changed in place is a new light curve.  The
        digest is taken over the arrays' own buffers where they are float64 and contiguous (numpy arrays, pandas Series as
        the reference driver passes them: code/synthetic_datasets/synth_mcmc.py:170-172) -- a repeated call costs three
        checksums, no copy (round 5: the conversions and the hash of copies were a third of the host entry's Python time)."""
        x, y, yerr = _as_f64(x), _as_f64(y), _as_f64(yerr)
        dig = (x.size, y.size, yerr.size, zlib.crc32(x), zlib.adler32(x), zlib.crc32(y), zlib.adler32(y), zlib.crc32(yerr), zlib.adler32(yerr))
        slot = self._slots.get(dig)
        if slot is None:
            if len(self._order) >= _capi.MAX_DATASETS:
                old = self._order.pop(0)
                slot = self._slots.pop(old)
            else:
                slot = len(self._order)
            self.handle.set_dataset(slot, x, y, yerr)
            self._slots[dig] = slot
        else:
            self._order.remove(dig)
        self._order.append(dig)
        return slot

    def set_prior(self, lower, upper, log_mask):
        key = (None if lower is None else (np.asarray(lower, float).tobytes(), np.asarray(upper, float).tobytes()), int(log_mask))
        if key != self._prior_key:
            self.handle.set_prior(lower, upper, log_mask)
            self._prior_key = key


def engine(cfg, GRBtype=None, device=-1, _pin=False):
    """Cached Engine for (model configuration, grid, device).  The least recently used engines beyond MAX_ENGINES are
    closed, except those a caller currently holds through use().  Front ends that evaluate go through use(); a bare
    engine() is a look-up for single-threaded callers (tests, introspection)."""
    key = (cfg.__dict__.get("_mp_key") or _cfg_key(cfg), "S" if GRBtype == "S" else "L", tuple(int(d) for d in device) if np.ndim(device) > 0 else int(device))   # (_mp_key: set by the front ends on configurations they cache and never modify)
    with _lock:
        e = _handles.pop(key, None)
        if e is None:
            for k in [k for k, v in _handles.items() if v.pins == 0][:max(0, len(_handles) - MAX_ENGINES + 1)]:
                _handles.pop(k).handle.close()       # nobody is inside an unpinned engine
            e = Engine(cfg, grid(GRBtype), device)
        _handles[key] = e                            # most recently used last
        if _pin:
            e.pins += 1
        return e


def acquire(cfg, GRBtype=None, device=-1):
    """The cached Engine, pinned against eviction and with its lock held; pair with release() (what `use` does)."""
    while True:
        e = engine(cfg, GRBtype, device, _pin=True)  # look-up (or creation) and pin in one critical section
        e.lock.acquire()
        if getattr(e.handle, "_h", True) is not None:
            return e
        e.lock.release()                             # closed by clear() while this thread waited: take a fresh one
        with _lock:
            e.pins -= 1


def release(e):
    e.lock.release()
    with _lock:
        e.pins -= 1


@contextlib.contextmanager
def use(cfg, GRBtype=None, device=-1):
    """`with engine.use(cfg, ...) as eng:` — the cached Engine, pinned against eviction and with its lock held for the
    duration of the block (the pin is taken under the module lock, so another thread's engine() for a ninth
    configuration can never close the handle between look-up and use; the module lock is not held while waiting)."""
    e = acquire(cfg, GRBtype, device)
    try:
        yield e
    finally:
        release(e)


def clear():
    """Close every cached engine.  An engine that a thread is using (inside `with use(...)`) is closed only after that
    thread has left its block: the cache forgets all engines at once, under the module lock, and then each one's own lock is
    taken before its handle is destroyed (use() holds that lock for the whole block; mp_destroy on a handle in use would be a
    use-after-free on the native side).  A thread that was still waiting for the engine's lock finds it closed and looks up a
    fresh one (use() above).  The engines' locks are re-entrant: a thread that calls clear() INSIDE its own `with use(...)`
    block (a cleanup helper, a test fixture) does not wait for itself -- its engine is closed then and there, and what the
    block does with it afterwards raises instead of touching freed memory."""
    with _lock:
        engines = list(_handles.values())
        _handles.clear()
    for e in engines:
        with e.lock:
            e.handle.close()
