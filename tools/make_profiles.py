"""Assemble profiles/r02_kernel_stats_large_beam5_b{1,8}_eager.md from the outputs of tools/run_r2z.sh (kernel statistics of the
final build) and tools/profile_r02b.sh (per-grid table), both merged under gpurun_out/ by gpurun.

    python tools/make_profiles.py
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEAD = {
    1: "# rocprofv3 --kernel-trace --stats, round 2 final state, Whisper large-v2 beam 5, 3.84 s clip, one utterance per device batch (the headline configuration): decode rows M = 5",
    8: "# rocprofv3 --kernel-trace --stats, round 2 final state, Whisper large-v2 beam 5, 3.84 s clip, 8 utterances per device batch (BASELINE configs[3] shape on one GPU): decode rows M = 40, the fragment-image path",
}
BODY = """
Command (GPU box, `tools/run_r2z.sh`): `WIS_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -d ... -- python bench.py --steps 5 --warmup 2 --batch {B} --no-cpu-baseline --no-extras`
(eager launches because rocprofv3 crashes inside HIP-graph capture; 7 generate calls + 6 roofline-tap passes over the decoder weight stream).
Aggregated from the rocpd sqlite output (`kernels` view) with `tools/prof_summary.py`.  Kernel durations in a dependent chain are INCLUSIVE of the boundary
before them (consecutive kernels show a gap of 0.0 us in the trace: start(k+1) = end(k)), so the column sums to the wall time of the chain.

Template arguments: `gemv_kernel<MB, MODE, SC, RM, W8>` (MODE 1 = LayerNorm-folded projection on raw fp32 rows, MODE 2 = f16 activations; SC = compile-time k-steps per wave, 0 = generic ring: FFN2),
`gemv_dual_kernel<SCA, SCB>` (out-projection + folded cross-Q in one launch), `gemv_frag_kernel<MB, PF, W8>` (batched rows on fragment images), `dec_cross_attn_kernel<TPW, CM, FOLD>`,
`gemm_f16_kernel<Epi, BM, BN, WM, WN>` / `gemm_pp_kernel<Epi>` (encoder GEMM tiles; the ping-pong 256x128 workgroup), `enc_attn_kernel<SPLIT>` (SPLIT = two workgroups per query tile and head),
`splitk_reduce_ln_kernel<SPLITS>` (FFN2 reduction + the next LayerNorm), `layernorm_kernel<AFFINE>`.

```
{stats}```
"""
GRID = """
Per kernel and grid size (threads), from the run one commit earlier (`tools/profile_r02b.sh`; same kernels, accumulators still in AGPRs): the skinny GEMMs by matrix
(20480 = 80 tiles: d x d or FFN2; 61440 = QKV; 81920 = FFN1; 829952 = vocabulary projection)

```
{grid}```
"""


def main():
    for B in (1, 8):
        stats = open(os.path.join(ROOT, "gpurun_out", "r02z", f"kernel_stats_b{B}.txt")).read()
        out = HEAD[B] + "\n" + BODY.format(B=B, stats=stats)
        g = os.path.join(ROOT, "gpurun_out", "r02", f"kernels_by_grid_b{B}.txt")
        if os.path.exists(g):
            out += GRID.format(grid=open(g).read())
        with open(os.path.join(ROOT, "profiles", f"r02_kernel_stats_large_beam5_b{B}_eager.md"), "w") as f:
            f.write(out)
        print("wrote", f.name)


if __name__ == "__main__":
    main()
