#!/bin/bash
# Encoder GEMM tile exploration (GPU box): per (kernel, grid) average duration for each WIS_GEMM_TILE override.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WIS_NO_GRAPH=1
for T in ${GT_TILES:-auto 128x128 64x128 256x256}; do
  if [ $T = auto ]; then unset WIS_GEMM_TILE; else export WIS_GEMM_TILE=$T; fi
  rm -rf gpurun_out/gt_$T
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/gt_$T -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch ${GT_BATCH:-1} > gpurun_out/gt_$T.log 2>&1
  python - "$T" <<'PY'
import sqlite3, sys, glob, re
T = sys.argv[1]
db = glob.glob(f"gpurun_out/gt_{T}/*.db")[0]
c = sqlite3.connect(db)
rows = c.execute("select name, grid_x/workgroup_x, grid_z, count(*), avg(end-start)/1e3 from kernels where name like '%gemm_f16%' or name like '%splitk%' group by 1,2,3 order by 1,2").fetchall()
tot = 0
print(f"== tile {T}")
for n, g, z, cnt, avg in rows:
    nm = re.sub(r"\(.*", "", n).replace("void wis::", "")
    print(f"   {nm:48s} wgs={g:5d} z={z} n={cnt:4d} avg={avg:7.2f} us")
    if cnt >= 32: tot += avg * (cnt / 96.0 if cnt >= 96 else 1)
print(f"   per-layer sum of recurring GEMM kernels ~ {tot:.1f} us")
PY
done
