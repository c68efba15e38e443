"""Running observation normaliser, host object + device-state bridge.

API and pickled attribute names (``rs._n/_M/_S``, ``demean/destd/clip``) are those of
/root/reference/utils/zfilter.py:7-74 so checkpoints holding a ``running_state`` stay loadable and
``running_state.rs.mean/std/n`` keep working (ego_mimic_eval.py:67,126). The lockstep rollout does not
call this object per sample: it moves the statistics to HBM (``to_device_state``), updates them with
the batched Chan merge kernel (K6) and writes them back (``from_device_state``).
"""
from __future__ import annotations

import numpy as np


class RunningStat:
    def __init__(self, shape):
        self._n = 0
        self._M = np.zeros(shape)
        self._S = np.zeros(shape)

    def push(self, x):
        """One Welford step (used by single-env callers such as the eval scripts)."""
        x = np.asarray(x, dtype=float)
        if x.shape != self._M.shape:
            raise AssertionError("RunningStat.push: shape %s != %s" % (x.shape, self._M.shape))
        self._n += 1
        delta = x - self._M
        if self._n == 1:
            self._M = self._M + delta
        else:
            self._M = self._M + delta / self._n
            self._S = self._S + delta * (x - self._M)

    def merge(self, count, mean, m2):
        """Chan merge of an already reduced block (count, mean, sum of squared deviations)."""
        count = int(count)
        if count == 0:
            return
        if self._n == 0:
            self._n, self._M, self._S = count, np.array(mean, float), np.array(m2, float)
            return
        tot = self._n + count
        d = np.asarray(mean, float) - self._M
        self._S = self._S + np.asarray(m2, float) + d * d * (self._n * count / tot)
        self._M = self._M + d * (count / tot)
        self._n = tot

    n = property(lambda self: self._n)
    mean = property(lambda self: self._M)
    shape = property(lambda self: self._M.shape)

    @property
    def var(self):
        return self._S / (self._n - 1) if self._n > 1 else np.square(self._M)

    @property
    def std(self):
        return np.sqrt(self.var)


class ZFilter:
    """y = clip((x - mean) / (std + 1e-8), +-clip) with running mean/std."""

    def __init__(self, shape, demean=True, destd=True, clip=10.0):
        self.demean, self.destd, self.clip = demean, destd, clip
        self.rs = RunningStat(shape)

    def __call__(self, x, update=True):
        if update:
            self.rs.push(x)
        y = np.asarray(x, dtype=float)
        if self.demean:
            y = y - self.rs.mean
        if self.destd:
            y = y / (self.rs.std + 1e-8)
        if self.clip:
            y = np.clip(y, -self.clip, self.clip)
        return y

    def set_mean_std(self, mean, std, n):
        # the reference stores `std` into S verbatim (utils/zfilter.py:69-74); kept for compatibility
        self.rs._n = n
        self.rs._M[...] = mean
        self.rs._S[...] = std

    # ------------------------------------------------------------------ device bridge (K6 state layout)
    def to_device_state(self, device):
        import torch
        d = int(np.prod(self.rs.shape))
        st = np.empty(1 + 2 * d)
        st[0] = self.rs._n
        st[1:1 + d] = np.asarray(self.rs._M, float).ravel()
        st[1 + d:] = np.asarray(self.rs._S, float).ravel()
        return torch.as_tensor(st, dtype=torch.float64, device=device)

    def from_device_state(self, state):
        st = state.detach().cpu().numpy()
        d = int(np.prod(self.rs.shape))
        self.rs._n = int(round(st[0]))
        self.rs._M = st[1:1 + d].reshape(self.rs.shape).copy()
        self.rs._S = st[1 + d:].reshape(self.rs.shape).copy()


# ---------------------------------------------------------------------- checkpoint compatibility
import contextlib
import sys
import types

_REF_MODULE = "utils.zfilter"      # where the reference defines RunningStat / ZFilter (utils/zfilter.py)


import pickle
import threading

_NAMES_LOCK = threading.RLock()


class _RefPickler(pickle._Pickler):
    """pickle.Pickler that writes RunningStat / ZFilter under the reference's module path, `utils.zfilter` -- per pickler, no
    process-wide state touched (the pure-Python pickler: the C one offers no hook for a class's global name)."""

    def save_global(self, obj, name=None):
        if obj is RunningStat or obj is ZFilter:
            self.write(pickle.GLOBAL + _REF_MODULE.encode() + b"\n" + obj.__name__.encode() + b"\n")
            self.memoize(obj)
            return
        super().save_global(obj, name)


class _RefUnpickler(pickle.Unpickler):
    """pickle.Unpickler that resolves `utils.zfilter.{RunningStat, ZFilter}` to this module's classes, whatever `utils` package
    (if any) is importable in the process."""

    def find_class(self, module, name):
        if module == _REF_MODULE and name in ("RunningStat", "ZFilter"):
            return {"RunningStat": RunningStat, "ZFilter": ZFilter}[name]
        return super().find_class(module, name)


def dump_reference_pickle(obj, f):
    """pickle.dump(obj, f) with `running_state` objects named as the reference names them (ego_pose/ego_mimic.py:135-139): the
    file loads in the unmodified reference. Thread-safe (nothing global changes); what Trainer.save uses."""
    _RefPickler(f, protocol=pickle.DEFAULT_PROTOCOL).dump(obj)


def load_reference_pickle(f):
    """pickle.load(f) for checkpoints written by the reference or by dump_reference_pickle. Thread-safe."""
    return _RefUnpickler(f).load()


@contextlib.contextmanager
def reference_pickle_names():
    """Context-manager form for code that calls plain `pickle.dump / pickle.load` itself. PROCESS-WIDE while active (it renames
    the classes and may install alias modules), serialised by a lock -- prefer dump_reference_pickle / load_reference_pickle.
    While active, RunningStat / ZFilter pickle under the reference's module path (`utils.zfilter.ZFilter`), so a
    checkpoint's `running_state` (ego_pose/ego_mimic.py:135-139) written here loads in the unmodified reference and
    one written by the reference loads here -- whether or not `egopose_amd/compat` is on the path. When `utils.zfilter`
    is not importable in this process, alias modules stand in for the duration of the block and are removed afterwards."""
    _NAMES_LOCK.acquire()
    classes = (RunningStat, ZFilter)
    saved = [c.__module__ for c in classes]
    installed = []
    cur = sys.modules.get(_REF_MODULE)
    if cur is None or getattr(cur, "ZFilter", None) is not ZFilter:
        try:
            import importlib
            cur = importlib.import_module(_REF_MODULE)           # egopose_amd/compat on the path: its re-export
        except Exception:
            cur = None
        if cur is None or getattr(cur, "ZFilter", None) is not ZFilter:
            prev = {k: sys.modules.get(k) for k in ("utils", _REF_MODULE)}
            alias = types.ModuleType(_REF_MODULE)
            alias.RunningStat, alias.ZFilter = RunningStat, ZFilter
            pkg = prev["utils"] if prev["utils"] is not None else types.ModuleType("utils")
            if prev["utils"] is None:
                pkg.__path__ = []
                sys.modules["utils"] = pkg
            sys.modules[_REF_MODULE] = alias
            installed = [prev]
    for c in classes:
        c.__module__ = _REF_MODULE
    try:
        yield
    finally:
        for c, m in zip(classes, saved):
            c.__module__ = m
        for prev in installed:
            for k, v in prev.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v
        _NAMES_LOCK.release()
