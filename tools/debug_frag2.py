"""Round-3 reproducer of the two-n-tile skinny GEMM's wrong tiles (csrc/dec_kernels.hip gemv_frag2_kernel; cause found in round 4: SLP-packed f32
epilogue arithmetic at three waves per SIMD, on every box the SLP build ran on - DESIGN section 4; the kernel is on by default since and this script is clean): LayerNorm-folded
projection at 80 rows x 51872 columns through wis_op_gemv, twice in one process, then neighbouring shapes; prints where the result
leaves the fp64 reference by more than 0.05 (tile index, column within the tile, row blocks).
    WIS_FRAG_NB=2 python tools/debug_frag2.py        # DBG48=1: four launches at 48 rows instead
"""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "willow-inference-server_amd"))
from wis_hip import _lib
from wis_hip._lib import DevBuf, check
lib = _lib.load()
def run(M, N, K, flags):
    rng = np.random.default_rng(M * 31 + N)
    x = rng.standard_normal((M, K)).astype(np.float32) * 2 + 0.5
    Wt = (rng.standard_normal((N, K)) * 0.05).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32); b = (0.1 * rng.standard_normal(K)).astype(np.float32)
    mu = x.astype(np.float64).mean(1, keepdims=True); var = x.astype(np.float64).var(1, keepdims=True)
    Wg = (Wt.astype(np.float32) * g).astype(np.float16).astype(np.float64)
    ref = ((x.astype(np.float64) - mu) / np.sqrt(var + 1e-5)) @ Wg.T + Wt.astype(np.float64) @ b.astype(np.float64) + bias
    dx, dW, dbias, dg, db = DevBuf.from_numpy(x), DevBuf.from_numpy(Wt), DevBuf.from_numpy(bias), DevBuf.from_numpy(g), DevBuf.from_numpy(b)
    dy = DevBuf(M * N * 4)
    check(lib.wis_op_gemv(0, dx.ptr, dg.ptr, db.ptr, dW.ptr, dbias.ptr, dy.ptr, M, N, K, flags))
    out = dy.to_numpy(np.float32, (M, N)).astype(np.float64)
    err = np.abs(out - ref)
    bad = np.argwhere(err > 0.05)
    print(f"M{M} N{N}: max err {err.max():.3e}, entries > 0.05: {len(bad)}")
    if len(bad):
        rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
        print("  rows:", rows[:40], "n rows", len(rows))
        print("  cols (first 40):", cols[:40], "n cols", len(cols), "col//16 unique", np.unique(cols // 16)[:30], "col%32//16:", np.unique((cols % 32) // 16))
        print("  sample:", [(int(r), int(c), float(out[r, c]), float(ref[r, c])) for r, c in bad[:6]])
import os
for M, N in ([(48, 51872)] * 4 if os.environ.get('DBG48') else ((80, 51872), (80, 51872), (72, 51872), (80, 25600), (96, 51872))):
    run(M, N, 1280, 12)
