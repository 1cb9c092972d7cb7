"""Where does the avg / min spread of a decode-step kernel come from?  python tools/spread.py <rocpd db> [layers]
For each skinny-GEMM / attention kernel of the one-utterance step (one launch per decoder layer and step) the launches are taken in
time order and folded to [steps][layers]: a layer position that is slow in EVERY step points at addresses (weights of that layer: HBM
channel / TLB reach), a step that is slow at every layer at clock state, and neither at contention with what ran before."""
import sqlite3, sys
import numpy as np
db, L = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 32
c = sqlite3.connect(db)
rows = c.execute("select name, grid_x, start, end from kernels order by start").fetchall()
groups = {}
for name, grid, s, e in rows:
    if not any(k in name for k in ("gemv_kernel", "gemv_dual", "dec_cross_attn", "dec_self_attn")):
        continue
    groups.setdefault((name.split("(")[0].replace("void ", "")[-60:], grid), []).append((e - s) / 1e3)
for (name, grid), d in sorted(groups.items(), key=lambda kv: -len(kv[1])):
    d = np.array(d)
    n = (len(d) // L) * L
    if n < 4 * L:
        continue
    m = d[-n:].reshape(-1, L)
    lay, stp = m.mean(0), m.mean(1)
    resid = m - lay[None, :] - stp[:, None] + m.mean()
    print(f"{name} grid {grid}: n={len(d)} mean {d.mean():.2f} min {d.min():.2f} p50 {np.median(d):.2f} p90 {np.percentile(d, 90):.2f} us | "
          f"std across layer positions {lay.std():.2f} (min {lay.min():.2f} max {lay.max():.2f}), across steps {stp.std():.2f}, residual {resid.std():.2f}")
