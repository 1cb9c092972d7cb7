#!/bin/bash
# Same-box A/B of bench.py between source trees (boxes of the pool differ by 1.5-3 %, more than most single changes):
#   gpurun -- 'bash tools/ab_bench.sh <dir>:<tag>[:ENV=VAL] ...'      e.g.  _ab/r2:r2 .:head .:head_nospin:WIS_CA_SPIN=0
# <dir> = a built tree (git archive <rev> | tar -x -C _ab/<name>; python <dir>/willow-inference-server_amd/build.py; _ab/ is
# git-ignored but travels with the snapshot).  Prints p50 ms, decode ms, encoder ms of the one-utterance headline per spec.
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/${WIS_TAG:-r3k}; mkdir -p $O; R=$GRAFT_REPO_ROOT
one() { # dir tag envs
  local d=$1 t=$2; shift 2
  ( cd $d && env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --clip 3sec --no-cpu-baseline --no-extras --no-roofline > $O/ab_${t}.json 2> $O/ab_${t}.err )
  python - $O/ab_${t}.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], d['p50_ms'], d['stage_ms_last_step']['decode_ms'], d['stage_ms_last_step']['encoder_ms'])
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
}
for spec in "$@"; do
  IFS=: read d t e <<< "$spec"
  one $d $t ${e:-A=1}
done
