"""Tuning aid: print the phase stamps (shader clock cycles) of decoder layer 0 kernels (needs a tap build:
WIS_EXTRA_HIPFLAGS=-DWIS_TAPS=1 python willow-inference-server_amd/build.py)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "willow-inference-server_amd"))
from wis_hip import _lib, ctranslate2 as ct2, weights as W

size = sys.argv[1] if len(sys.argv) > 1 else "large"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = _lib.load()
a = W.arch(size)
w = W.synthetic_weights(size)
arena, index = W.build_arena(w)
h = ct2.create_handle(a, arena, index, 0, max_batch=B, max_beam=5)
out = np.zeros((6, 16), np.uint64)
names = ["gemv QKV (LN)", "gemv out-proj", "cross-attn", "self-attn", "gemv FFN1 (LN)", "gemv FFN2"]
if B * 5 > 8:      # the batched-row kernels (round 6): gemv_frag_kernel<3, 10> QKV, gemv_frag3_kernel (out-projection + the two q halves), -, gemv_frag_kernel d x d,
    # gemv_frag2_kernel (FFN1, two n-tiles per workgroup), gemv_frag_kernel<3, 8> with two K slices (FFN2)
    names = ["frag QKV (LN) 240 wg", "frag3 out-proj + q halves 240 wg", "cross-attn", "frag cross-out d x d 80 wg", "frag2 FFN1 (LN) 160 wg", "frag FFN2 2 K slices 160 wg"]
# gemv_body's stamps (csrc/dec_kernels.hip): one wave of workgroup 0
PH = ["requests issued (activation rows + first weight fragments)", "activation rows arrived, cast to f16, staged in LDS", "staging barrier",
      "weight stream consumed (MFMA loop)", "epilogue operands + cross-wave reduction barrier", "epilogue, stores issued"]
if B * 5 > 8:
    PH = ["first PF k-steps requested (weight fragment + MB activation fragments each)", "stream consumed (MFMA loop, refills in flight)",
          "epilogue operands requested, partial sums to LDS, reduction barrier", "K-split: slice sums published, ticket (FFN2 only)", "epilogue (LayerNorm fold / residual / partials), stores issued",
          "tail"]
md = "--md" in sys.argv
for pos in (10,):
    _lib.check(lib.wis_debug_phase_cycles(h, B, 5, pos, out.ctypes.data_as(C.POINTER(C.c_uint64))))
    for i, n in enumerate(names):
        st = [int(v) for v in out[i][:14] if v]
        if n == "cross-attn":      # entries 0-5: workgroup (chunk 0, head 0, utterance 0) - dec_cross_attn_rs_kernel: its first K wave; 7-13: the first V wave of that workgroup
            # (dec_cross_attn_kernel, WIS_CA_RS=0: the last chunk's workgroup, the combiner of the granule form)
            comb = [int(v) for v in out[i][7:14] if v]
            st = [int(v) for v in out[i][:6] if v]
            if comb:
                print(f"{'cross-attn V wave (WIS_CA_RS=0: combiner)':16s} pos {pos}: total {comb[-1] - comb[0]:6d} cyc; phases {[comb[j + 1] - comb[j] for j in range(len(comb) - 1)]}; starts {comb[0] - st[0]:+d} after the producer")
        if st:
            ph = [st[j + 1] - st[j] for j in range(len(st) - 1)]
            print(f"{n:16s} pos {pos}: total {st[-1] - st[0]:6d} cyc; phases {ph}")
            if md and (n.startswith("gemv") or n.startswith("frag")) and len(ph) == len(PH):
                print(f"\n| {n}: phase (one wave of workgroup 0, decoder layer 0, {B} x beam 5 rows) | s_memtime ticks | share |\n|---|---|---|")
                for name, c in zip(PH, ph):
                    print(f"| {name} | {c} | {100.0 * c / max(1, st[-1] - st[0]):.0f} % |")
                print(f"| total | {st[-1] - st[0]} | |\n")
