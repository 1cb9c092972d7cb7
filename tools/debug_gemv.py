import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "willow-inference-server_amd"))
from wis_hip import _lib
from wis_hip._lib import DevBuf, check
from scipy.special import erf
lib = _lib.load()
def run(M, N, K, flags, seed=0):
    rng = np.random.default_rng(M * 31 + N + K)
    Wt = (rng.standard_normal((N, K)) * 0.05).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32); b = (0.1 * rng.standard_normal(K)).astype(np.float32)
    if flags & 8:
        x = (rng.standard_normal((M, K)) * 2 + 0.3).astype(np.float32)
        mu = x.astype(np.float64).mean(1, keepdims=True); var = x.astype(np.float64).var(1, keepdims=True)
        xin = (((x - mu) / np.sqrt(var + 1e-5)) * g + b).astype(np.float16).astype(np.float64)
    else:
        x = rng.standard_normal((M, K)).astype(np.float16); xin = x.astype(np.float64)
    ref = xin @ Wt.astype(np.float64).T + bias
    if flags & 1: ref = 0.5 * ref * (1 + erf(ref / np.sqrt(2)))
    y0 = rng.standard_normal((M, N)).astype(np.float32)
    if flags & 2: ref = ref + y0
    f32 = bool(flags & 6)
    dx, dW, dbias, dg, db = [DevBuf.from_numpy(a) for a in (x, Wt, bias, g, b)]
    dy = DevBuf.from_numpy(y0) if f32 else DevBuf(M * N * 2)
    check(lib.wis_op_gemv(0, dx.ptr, dg.ptr, db.ptr, dW.ptr, dbias.ptr, dy.ptr, M, N, K, flags))
    out = dy.to_numpy(np.float32 if f32 else np.float16, (M, N)).astype(np.float64)
    err = np.abs(out - ref)
    rel = np.linalg.norm(out - ref) / np.linalg.norm(ref)
    bad = err > 0.02 * (1 + np.abs(ref))
    print(f"M{M} N{N} K{K} flags{flags}: rel {rel:.3e} max {err.max():.3e} bad {bad.sum()} rows_with_bad {sorted(set(np.where(bad)[0]))[:12]} cols%16 {sorted(set(np.where(bad)[1] % 16))[:16]} ncols {len(set(np.where(bad)[1]))}")
for cfg in [(20, 5120, 640, 9), (20, 5120, 1280, 9), (24, 5120, 1280, 9), (40, 5120, 1280, 9), (40, 7680, 1280, 9), (40, 1280, 1280, 9), (33, 5120, 1280, 12)]:
    for rep in range(4):
        run(*cfg)
