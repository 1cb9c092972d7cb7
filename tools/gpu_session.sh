#!/bin/bash
# The gpurun command file of the build sessions (one documented script instead of one file per call):
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <step> [<step> ...]'
# Steps (each writes under gpurun_out/<tag>/, tag = $WIS_TAG or "r3"):
#   lab [M ...]     tools/bin/gemm_lab at the given row counts (default 1500 3000 12000), stamps off, weights from HBM
#   tests <expr>    pytest -m gpu -k <expr> (quote the expression); "tests all" = the whole GPU suite
#   smoke           __graft_entry__.smoke()
#   bench<N> [ENV=VAL ...]   bench.py at N utterances per device batch (no extras, no CPU baseline); extra words are
#                   environment assignments for that run and become part of the output name
#   eot             bench.py --natural-eot: the search that ends on EOT against the fixed-length call of the same number of passes
#   stream          bench.py's streaming rows alone (large-v2, 30sec.flac)
#   benchfull       the default bench.py line (what the driver runs)
#   prof<N> [ENV=VAL ...]   rocprofv3 --kernel-trace --stats of the eager bench at N utterances per device batch -> kernel_stats_b*.txt (+ by-grid table)
#   frag2 [iters]   tools/bin/frag2_lab in its three flag variants (as the library / -fno-slp-vectorize / accumulators in AGPRs): the two-n-tile
#                   skinny GEMM against the shipped kernel, bit for bit, with LDS dump + hardware ids of a failing workgroup -> frag2_*.txt
#   engine          tools/bin/engine_lab: the persistent decode-layer skeleton against the graph-replayed launch chain (40 / 20 / 8 k-steps of
#                   weights per CU and stage) -> engine_lab.txt
#   dbgfrag2        tools/debug_frag2.py with WIS_FRAG_NB=2 (the round-3 reproducer of the two-tile kernel's wrong tiles) -> dbgfrag2.txt
#   pmc <tag> "<COUNTER ...>" [batch]   one rocprofv3 --pmc pass of the eager bench (default 8 utterances) -> pmc_<tag>_b<batch>.txt (per kernel and grid: launches, mean per launch)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${WIS_TAG:-r5}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$O"
R=$GRAFT_REPO_ROOT
run_bench() {   # $1 = batch, rest = env assignments
  local B=$1; shift
  local name="bench_b${B}"; for kv in "$@"; do name="${name}_${kv//[^A-Za-z0-9=]/}"; done
  env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --batch "$B" --no-cpu-baseline --no-extras > "$O/$name.json" 2> "$O/$name.err"
  python - "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms/step", d["ms_per_step"], "x_rt", d["value"], "stages", d["stage_ms_last_step"], "roofline", (d.get("roofline") or {}).get("frac"),
          "dec_step", ((d.get("roofline") or {}).get("decode_step") or {}).get("ms"), "enc_TF", ((d.get("roofline") or {}).get("encoder") or {}).get("achieved_TFLOPs"))
except Exception as e:
    print("bench output unreadable:", e)
PY
}
while [ $# -gt 0 ]; do
  step=$1; shift
  case $step in
    lab)
      Ms=""; while [ $# -gt 0 ] && [[ $1 =~ ^[0-9]+$ ]]; do Ms="$Ms $1"; shift; done
      [ -z "$Ms" ] && Ms="1500 3000 12000"
      for M in $Ms; do timeout 300 tools/bin/gemm_lab "$M" 0 1 > "$O/lab_M$M.txt" 2>&1; grep -v "^  stamps\|steady" "$O/lab_M$M.txt"; done ;;
    tests)
      expr=$1; shift
      if [ "$expr" = all ]; then timeout 1500 python -m pytest tests -q -x -m gpu > "$O/tests_all.log" 2>&1; echo "rc=$?" >> "$O/tests_all.log"; tail -5 "$O/tests_all.log"
      else timeout 1500 python -m pytest tests -q -x -m gpu -k "$expr" -s > "$O/tests_sel.log" 2>&1; echo "rc=$?" >> "$O/tests_sel.log"; grep -v "^$" "$O/tests_sel.log" | tail -60; fi ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; tail -2 "$O/smoke.log" ;;
    bench[0-9]*)
      B=${step#bench}; envs=(); while [ $# -gt 0 ] && [[ $1 == *=* ]]; do envs+=("$1"); shift; done
      run_bench "$B" "${envs[@]}" ;;
    eot)      # eot [ENV=VAL ...]: the natural-EOT row; extra words are environment assignments (part of the output name)
      envs=(); name="bench_eot"; while [ $# -gt 0 ] && [[ $1 == *=* ]]; do envs+=("$1"); name="${name}_${1//[^A-Za-z0-9=]/}"; shift; done
      env "${envs[@]}" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-roofline --natural-eot > "$O/$name.json" 2> "$O/$name.err"
      grep "eot-trace" "$O/$name.err" | tail -24
      python - "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "headline ms", d["p50_ms"])
    for k, v in (d.get("natural_eot") or {}).items():
        print("  ", k, v if not isinstance(v, dict) else {a: b for a, b in v.items() if a.split()[0] in ("natural_eot_p50_ms", "fixed_length_same_passes_p50_ms", "natural_over_fixed", "decoder_passes_needed", "decoder_passes_enqueued", "decode_ms_natural", "decode_ms_fixed")})
except Exception as e:
    print("bench output unreadable:", e)
PY
      ;;
    phases)   # tap build (lib/libwis_hip_taps.so: python willow-inference-server_amd/build.py --variant taps -DWIS_TAPS=1): phase stamps of layer 0's skinny GEMMs
      for b in 1 8; do WIS_LIB_PATH=$R/willow-inference-server_amd/lib/libwis_hip_taps.so timeout 300 python tools/phase_cycles.py large $b --md > "$O/phases_b$b.txt" 2>&1; cat "$O/phases_b$b.txt" | head -60; done ;;
    encstress)   # encstress [rounds]: tools/enc_stress.py (bitwise repeatability of encoder + generate under 4-replica concurrency)
      RN=100; if [ $# -gt 0 ] && [[ $1 =~ ^[0-9]+$ ]]; then RN=$1; shift; fi
      timeout 900 python tools/enc_stress.py $RN > "$O/enc_stress_$RN.txt" 2>&1; echo "rc=$?" >> "$O/enc_stress_$RN.txt"; tail -5 "$O/enc_stress_$RN.txt" ;;
    stream)   # bench.py's streaming rows alone (BASELINE configs[4]): large-v2, 30sec.flac, beam 3 / beam 1 with speculation
      timeout 900 python - > "$O/stream.json" 2> "$O/stream.err" <<'PY'
import json, os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "willow-inference-server_amd")]
import bench
from wis_hip import _lib, ctranslate2 as ct2, weights as W
lib = _lib.load(); _lib.require_gpu()
a = W.arch("large"); w = W.synthetic_weights("large", seed=1234)
arena, index = W.build_arena(w)
h = ct2.create_handle(a, arena, index, 0, max_batch=8, max_beam=5)
print(json.dumps(bench.streaming_bench(h, a, 0, os.path.join("tests", "golden", "clips", "30sec.flac"))))
PY
      python -c "import json,sys; d=json.load(open('$O/stream.json')); print({k:v for k,v in d.items() if k!='noise_64s'})" || tail -5 "$O/stream.err" ;;
    benchfull) timeout 900 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; tail -c 600 "$O/bench_default.json" ;;
    prof[0-9]*)
      B=${step#prof}; while [ $# -gt 0 ] && [[ $1 == *=* ]]; do export "$1"; B="$B"; PSUF="${PSUF}_${1//[^A-Za-z0-9=]/}"; shift; done
      ( cd /tmp && export TMPDIR=/tmp && WIS_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_b$B" -o "b$B" -- python "$R/bench.py" --steps 5 --warmup 2 --batch "$B" --no-cpu-baseline --no-extras > "$O/bench_eager_b$B.log" 2>&1 )
      DB=$(find "$O/prof_b$B" -name "*.db" | head -1)
      python tools/prof_summary.py "$DB" 45 > "$O/kernel_stats_b$B$PSUF.txt" 2>&1
      python tools/prof_summary.py "$DB" 45 --by-grid > "$O/kernels_by_grid_b$B$PSUF.txt" 2>&1
      [ "$B" = 1 ] && python tools/spread.py "$DB" 32 > "$O/spread_b1.txt" 2>&1
      find "$O/prof_b$B" -name "*.db" -delete
      head -30 "$O/kernel_stats_b$B$PSUF.txt"; PSUF="" ;;
    frag2)
      IT=100; if [ $# -gt 0 ] && [[ $1 =~ ^[0-9]+$ ]]; then IT=$1; shift; fi
      for v in frag2_lab frag2_lab_noslp frag2_lab_agpr; do [ -x tools/bin/$v ] && { timeout 120 tools/bin/$v "$IT" > "$O/$v.txt" 2>&1; echo "== $v"; grep -c "differing words" "$O/$v.txt"; grep "^variant\|^shipped\|reference launch" "$O/$v.txt"; }; done ;;
    engine)
      : > "$O/engine_lab.txt"
      for ks in 40 20 8; do timeout 90 tools/bin/engine_lab 56 $ks 20 >> "$O/engine_lab.txt" 2>&1; echo "rc=$?" >> "$O/engine_lab.txt"; done
      cat "$O/engine_lab.txt" ;;
    dbgfrag2)
      WIS_FRAG_NB=2 timeout 300 python tools/debug_frag2.py > "$O/dbgfrag2.txt" 2>&1; echo "rc=$?" >> "$O/dbgfrag2.txt"; cat "$O/dbgfrag2.txt" | head -40 ;;
    pmc)      # pmc <tag> "<COUNTER ...>" [batch]: ONE rocprofv3 --pmc pass (counters only with --kernel-trace, as gpurun requires) of the eager bench
      PT=$1; CNT=$2; shift 2; PB=8; if [ $# -gt 0 ] && [[ $1 =~ ^[0-9]+$ ]]; then PB=$1; shift; fi
      ( cd /tmp && export TMPDIR=/tmp && WIS_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $CNT --kernel-trace -d "$O/pmc_$PT" -o p --output-format csv -- python "$R/bench.py" --steps 2 --warmup 1 --batch "$PB" --no-cpu-baseline --no-extras --no-roofline > "$O/pmc_$PT.log" 2>&1 )
      python tools/pmc_sum.py "$O/pmc_$PT" "${CNT// /,}" > "$O/pmc_${PT}_b$PB.txt" 2>&1; rm -rf "$O/pmc_$PT"; head -40 "$O/pmc_${PT}_b$PB.txt" ;;
    *) echo "unknown step $step" ;;
  esac
done
