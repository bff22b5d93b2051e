"""Deferred results for the drop-in API.

The reference harness (test_pMCTF_flex.py:196-258) calls `encode_one_stage` once per frame pair and only stores what
comes back: tensors go into `frames_coded`, bit counts into per-frame lists, the motion context into the next call.  None
of it is looked at before the next temporal stage starts.  The MI355X build uses that: `encode_one_stage` may hand back
DEFERRED values and collect the pairs of a stage, which are then coded as one batch the moment any value is needed —
eight times larger launches for the very same files, bits and tensors (pMCTF.encode_stage_pairs).

  * Deferred        — a value that is computed on first use.  Arithmetic on it gives another Deferred (so that
                      `curr_bits = r["bit_H"] + r["bit_ME"]; bpp = curr_bits / pixels` keeps deferring); anything that
                      needs the number — float(), comparison, formatting, indexing — forces it.
  * DeferredTensor  — the same for tensors: any torch function or attribute access forces it; the model's own entry points
                      unwrap it without a copy.
Forcing is transparent: a caller that inspects every result immediately simply gets the eager pair-by-pair schedule.
"""
import operator

import torch


class Deferred:
    __slots__ = ("_thunk", "_value", "_done", "_ready")

    def __init__(self, thunk, ready=None):
        self._thunk = thunk
        self._value = None
        self._done = False
        self._ready = ready         # optional: tells whether forcing would be free (the producer has already run)

    def force(self):
        if not self._done:
            v = self._thunk()
            while isinstance(v, Deferred):         # a deferred result of a deferred producer
                v = v.force()
            self._value = v
            self._done = True
            self._thunk = self._ready = None
        return self._value

    # ---- numbers: keep deferring through arithmetic
    def _bin(self, other, op, swap=False):
        cls = Deferred
        if swap:
            return cls(lambda: op(force(other), self.force()))
        return cls(lambda: op(self.force(), force(other)))

    def __add__(self, o): return self._bin(o, operator.add)
    def __radd__(self, o): return self._bin(o, operator.add, True)
    def __sub__(self, o): return self._bin(o, operator.sub)
    def __rsub__(self, o): return self._bin(o, operator.sub, True)
    def __mul__(self, o): return self._bin(o, operator.mul)
    def __rmul__(self, o): return self._bin(o, operator.mul, True)
    def __truediv__(self, o): return self._bin(o, operator.truediv)
    def __rtruediv__(self, o): return self._bin(o, operator.truediv, True)
    def __neg__(self): return Deferred(lambda: -self.force())

    # ---- everything else needs the value
    def __float__(self): return float(self.force())
    def __int__(self): return int(self.force())
    def __index__(self): return operator.index(self.force())
    def __bool__(self): return bool(self.force())
    def __round__(self, n=None): return round(self.force(), n)
    def __abs__(self): return abs(self.force())
    def __eq__(self, o): return self.force() == force(o)
    def __ne__(self, o): return self.force() != force(o)
    def __lt__(self, o): return self.force() < force(o)
    def __le__(self, o): return self.force() <= force(o)
    def __gt__(self, o): return self.force() > force(o)
    def __ge__(self, o): return self.force() >= force(o)
    def __hash__(self): return id(self)
    def __repr__(self): return repr(self.force())
    def __str__(self): return str(self.force())
    def __format__(self, spec): return format(self.force(), spec)
    def __getitem__(self, k): return self.force()[k]
    def __len__(self): return len(self.force())
    def __iter__(self): return iter(self.force())
    def __contains__(self, k): return k in self.force()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.force(), name)


class DeferredTensor(Deferred):
    __slots__ = ()

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return func(*unwrap(args), **unwrap(kwargs or {}))

    # tensor arithmetic is not deferred: it needs the data
    def _bin(self, other, op, swap=False):
        return op(force(other), self.force()) if swap else op(self.force(), force(other))

    def __neg__(self): return -self.force()

    def __array__(self, *args, **kwargs):
        return self.force().__array__(*args, **kwargs)


def force(x):
    return x.force() if isinstance(x, Deferred) else x


def unwrap(x):
    """force every Deferred inside nested lists / tuples / dicts"""
    if isinstance(x, Deferred):
        return x.force()
    if isinstance(x, (list, tuple)):
        return type(x)(unwrap(v) for v in x)
    if isinstance(x, dict):
        return {k: unwrap(v) for k, v in x.items()}
    return x


def is_pending(x):
    """a Deferred whose producer has not run yet"""
    return isinstance(x, Deferred) and not x._done and not (x._ready is not None and x._ready())
