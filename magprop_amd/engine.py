"""Handle / dataset caches shared by the reference-shaped front ends (funcs, mcmc_eqns, synth, LogProb)."""
import contextlib
import hashlib
import threading

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


def _cfg_key(cfg):
    return tuple(getattr(cfg, f[0]) for f in cfg._fields_)


class Engine:
    """One mp_handle plus a content-addressed cache of the datasets registered on it."""

    def __init__(self, cfg, tgrid, device=-1):
        self.handle = _capi.Handle(cfg, tgrid, device)
        self._slots = {}      # digest -> slot
        self._order = []      # LRU of digests
        self._prior_key = None
        self.lock = threading.Lock()
        self.pins = 0         # callers between use()'s entry and exit (guarded by the module lock): never evicted

    def dataset_slot(self, x, y, yerr):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        yerr = np.ascontiguousarray(yerr, dtype=np.float64)
        dig = hashlib.blake2b(x.tobytes() + y.tobytes() + yerr.tobytes(), digest_size=16).digest()
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
        key = (None if lower is None else (tuple(np.asarray(lower, float)), tuple(np.asarray(upper, float))),
               int(log_mask))
        if key != self._prior_key:
            self.handle.set_prior(lower, upper, log_mask)
            self._prior_key = key


def engine(cfg, GRBtype=None, device=-1, _pin=False):
    """Cached Engine for (model configuration, grid, device).  The least recently used engines beyond MAX_ENGINES are
    closed, except those a caller currently holds through use().  Front ends that evaluate go through use(); a bare
    engine() is a look-up for single-threaded callers (tests, introspection)."""
    key = (_cfg_key(cfg), "S" if GRBtype == "S" else "L", int(device))
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


@contextlib.contextmanager
def use(cfg, GRBtype=None, device=-1):
    """`with engine.use(cfg, ...) as eng:` — the cached Engine, pinned against eviction and with its lock held for the
    duration of the block (the pin is taken under the module lock, so another thread's engine() for a ninth
    configuration can never close the handle between look-up and use; the module lock is not held while waiting)."""
    while True:
        e = engine(cfg, GRBtype, device, _pin=True)  # look-up (or creation) and pin in one critical section
        e.lock.acquire()
        if getattr(e.handle, "_h", True) is not None:
            break
        e.lock.release()                             # closed by clear() while this thread waited: take a fresh one
        with _lock:
            e.pins -= 1
    try:
        yield e
    finally:
        e.lock.release()
        with _lock:
            e.pins -= 1


def clear():
    """Close every cached engine.  An engine that a thread is using (inside `with use(...)`) is closed only after that
    thread has left its block: the cache forgets all engines at once, under the module lock, and then each one's own lock is
    taken before its handle is destroyed (use() holds that lock for the whole block; mp_destroy on a handle in use would be a
    use-after-free on the native side).  A thread that was still waiting for the engine's lock finds it closed and looks up a
    fresh one (use() above)."""
    with _lock:
        engines = list(_handles.values())
        _handles.clear()
    for e in engines:
        with e.lock:
            e.handle.close()
