"""Device timeline of one decode step (wis_debug_timeline): per layer-kernel duration and the idle gap before it.
usage: python tools/timeline.py [size] [beam] [pos] [batch]   (needs a tap build: WIS_EXTRA_HIPFLAGS=-DWIS_TAPS=1 python willow-inference-server_amd/build.py)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "willow-inference-server_amd"))
from wis_hip import _lib, ctranslate2 as ct2  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "large"
beam = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pos = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
model = ct2.Whisper(f"synthetic:{size}", max_batch=B, max_beam=5)
L = model.arch["n_layers"]
names = ["QKV", "self-attn", "out", "cross-Q", "cross-attn", "cross-out", "FFN1", "FFN2"]
lib = _lib.load()
for graph in (1,):
    out = np.zeros((L * 8, 2), np.uint64)
    _lib.check(lib.wis_debug_timeline(model._replicas[0].handle, B, beam, pos, graph, out.ctypes.data_as(C.POINTER(C.c_uint64)), L * 8))
    t = out.astype(np.int64) * 10           # ns (100 MHz clock)
    dur = (t[:, 1] - t[:, 0]).reshape(L, 8)
    gap = np.zeros(L * 8, np.int64)
    gap[1:] = t[1:, 0] - t[:-1, 1]
    gap = gap.reshape(L, 8)
    span = t[-1, 1] - t[0, 0]
    print(f"== {'graph replay' if graph else 'eager'}: {L} layers, first start -> last end {span / 1e3:.1f} us = {span / 1e3 / (L * 8):.2f} us per kernel")
    print("   kernel        duration us (median over layers 1..)   gap before us (median)")
    for j, n in enumerate(names):
        print(f"   {n:12s}  {np.median(dur[1:, j]) / 1e3:8.2f}   [{dur[1:, j].min() / 1e3:.2f} .. {dur[1:, j].max() / 1e3:.2f}]     {np.median(gap[1:, j]) / 1e3:8.2f}")
    print(f"   sum of medians: durations {np.median(dur[1:], axis=0).sum() / 1e3:.2f} us/layer, gaps {np.median(gap[1:], axis=0).sum() / 1e3:.2f} us/layer")
