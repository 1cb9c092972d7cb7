"""Run the REAL reference `wis.audio.find_longest_common_sequence` under the array-comparison semantics of its pinned
numpy (1.23.5, /root/reference/requirements.txt:58) while the interpreter carries a newer numpy.

The only numpy behaviour that function depends on and that changed since is `np.array(a) == np.array(b)` for shapes that
do not broadcast ((2,) vs (3,), or (0,) vs (2,)): numpy < 1.25 emitted a DeprecationWarning and returned the scalar False
(so `np.sum(...)` is 0), numpy >= 1.25 raises ValueError.  Shapes that DO broadcast (a one-element running sequence against
a longer head) behave the same in both.  The shim swaps the module's `np` for a proxy whose `array()` returns an ndarray
subclass with exactly that fallback; everything else is the reference's own code, untouched."""
import contextlib
import types

import numpy as np


class _Arr123(np.ndarray):
    def __eq__(self, other):
        try:
            return np.ndarray.__eq__(self.view(np.ndarray), np.asarray(other))
        except ValueError:
            return np.False_

    __hash__ = None


def _array(*a, **k):
    return np.array(*a, **k).view(_Arr123)


@contextlib.contextmanager
def numpy_1_23_semantics(module):
    proxy = types.ModuleType("numpy_1_23_proxy")
    proxy.__dict__.update({k: getattr(np, k) for k in dir(np) if not k.startswith("__")})
    proxy.array = _array
    saved = module.np
    module.np = proxy
    try:
        yield
    finally:
        module.np = saved
