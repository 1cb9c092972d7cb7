#!/bin/bash
# PMC passes over the encoder kernels (run on the GPU box): tools/pmc_gemm.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WIS_NO_GRAPH=1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d gpurun_out/pmc_g$i -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_g$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for i in (1, 2):
    files = glob.glob(f"gpurun_out/pmc_g{i}/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no csv for pass", i); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0][:70]
        if not any(s in k for s in ("gemm_f16", "enc_attn", "gemv_kernel<1, 2, 0", "layernorm")):
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in acc.items():
        print(k)
        for c, x in sorted(v.items()):
            print(f"    {c:28s} {x:.4g}")
PY
