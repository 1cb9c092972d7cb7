"""Assemble the committed per-round artefacts (--round r04; default r03) under profiles/ from what tools/gpu_session.sh left under gpurun_out/<tag>/:

    python tools/make_profiles.py --stats r3A --pmc r3B --tcc r3s --lab r3c --lab-single r3b --bench r3A

  r03_kernel_stats_large_beam5_b{1,8,16}_eager.md   rocprofv3 --kernel-trace --stats (+ per-grid table) of the eager bench
  r03_pmc_decode.json                               HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE) of the decoder kernels at 1 and 8 utterances
  r03_pmc_encoder_sq.md                             SQ counters of the encoder kernels at 8 utterances (MFMA busy, wait shares, LDS conflicts)
  r03_gemm_lab.md                                   tools/gemm_lab: the encoder GEMM loop variants incl. the 8-phase kernel and its ablations
  r03_bench_large_beam5.json                        the default bench.py line
"""
import argparse
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
RND = "r03"      # file prefix under profiles/ (--round)
P = os.path.join(ROOT, "profiles")
D = 1280
# (round 6, round-5 review item 7: the one-utterance kernels are matched by the LEADING template arguments - MB, MODE, SC = k-steps per wave - and the grid,
# so that a template parameter added behind them does not drop a kernel from the table: round 5's names carried six arguments, the keys four, and
# `roofline.traffic` came from gemv_dual_kernel alone)
ALG_B1 = {r"gemv_kernel<1, 2, 40, .* 20480": 8 * D * D,            # FFN2: all 40 fragments of a wave up front
          r"gemv_kernel<1, 1, 20, .* 40960": 8 * D * D,            # FFN1 on two-tile workgroups
          r"gemv_kernel<1, 1, 10, .* 61440": 6 * D * D,            # QKV
          r"gemv_kernel<1, 2, 10, .* 20480": 2 * D * D,            # cross-attention output projection
          r"gemv_dual_kernel<10, 20> 40960": 6 * D * D,            # out-projection + the folded cross-Q
          r"gemv_kernel<1, 1, 20, .* 414976": 2 * 51872 * D,       # vocabulary projection (two-tile workgroups)
          r"dec_cross_attn_kernel 30720": 2 * 2 * 1500 * D,
          r"dec_cross_attn_rs_kernel 30720": 2 * 2 * 1500 * D}       # round 6: the role-split form (K waves / V waves)
ALG_B8 = {"gemv_frag_kernel<3, 8, false> 40960": 8 * D * D,      # FFN2: two K slices per n-tile (grid.y = 2)
          "gemv_frag_kernel<3, 10, false> 81920": 8 * D * D, "gemv_frag_kernel<3, 10, false> 61440": 6 * D * D,
          "gemv_frag_kernel<3, 10, false> 20480": 2 * D * D, "gemv_frag_kernel<3, 10, false> 829952": 2 * 51872 * D, "dec_cross_attn_kernel 245760": 8 * 2 * 2 * 1500 * D,
          # round 4: FFN1 and the vocabulary on two-tile workgroups (160 / 1621 workgroups), out-projection + the two halves of the folded cross-Q in one launch (240 workgroups)
          "gemv_frag2_kernel<3, 6, false> 40960": 8 * D * D, "gemv_frag2_kernel<3, 6, false> 414976": 2 * 51872 * D, "gemv_frag3_kernel<3, 10> 61440": 6 * D * D,
          # round 6: the role-split cross-attention; the d x d projection and FFN2 split by row blocks (80 n-tiles x 3 row-block groups = 240 workgroups)
          "dec_cross_attn_rs_kernel 245760": 8 * 2 * 2 * 1500 * D, "gemv_frag_ms_kernel<1, 10, false> 61440": 2 * D * D, "gemv_frag_ms_kernel<1, 8, false> 61440": 8 * D * D}
TEMPLATE_NOTE = ("Template arguments: `gemv_kernel<MB, MODE, SC, RM, W8>` (MODE 1 = LayerNorm-folded projection on raw fp32 rows, MODE 2 = f16 activations; SC = compile-time k-steps per wave, 0 = "
                 "generic ring: FFN2), `gemv_dual_kernel<SCA, SCB>` (out-projection + folded cross-Q in one launch), `gemv_frag_kernel<MB, PF, W8>` (batched rows on fragment images: MB 16-row blocks, "
                 "PF k-steps in flight), `gemv_frag_ms_kernel<MB, PF, W8>` (round 6: the same split by row blocks over workgroups - grid (n-tiles, 1, row-block groups): the d x d projection and FFN2), `dec_cross_attn_kernel<TPW, CM, FOLD, SPIN>` (SPIN = granule hand-off of the chunk partials), `dec_cross_attn_rs_kernel<SPIN, NT, NKW>` (round 6: K waves / V waves; the decode steps' cross-attention), `gemm_8p_kernel<Epi, TR>` (8-phase 256 x 256 LDS-DMA GEMM, "
                 "persistent over tiles; TR = swapped operands for the V images; EpiResid = bias + fp32 residual), `gemm_8pn_kernel<Epi>` (the same on a 128 x 256 tile: FFN1 of one utterance), `gemm_f16_kernel<Epi, BM, BN, WM, WN>` / `gemm_pp_kernel<Epi>` (register-staged tiles / ping-pong 256 x 128), `enc_attn_lazy_kernel<SPLIT>` (r4: lazy softmax reference; `enc_attn_kernel<SPLIT>` is the A/B form behind WIS_ENC_ATTN_LAZY=0), "
                 "`splitk_reduce_ln_kernel<SPLITS>`, `layernorm_kernel<AFFINE>`.  By-grid table: 20480 threads = 80 tiles (d x d; FFN2 at one utterance), 40960 = FFN2 of the batched path (two K slices), 61440 = QKV, 81920 = FFN1, 829952 = vocabulary projection.")


def read(path):
    return open(path).read() if os.path.exists(path) else None


def kernel_stats(tag):
    rows = {1: "one utterance per device batch (the headline configuration): decode rows M = 5", 8: "8 utterances per device batch (BASELINE configs[3] shape on one GPU): decode rows M = 40",
            16: "16 utterances per device batch: decode rows M = 80 (five row blocks)"}
    for B, what in rows.items():
        st, gr = read(f"{G}/{tag}/kernel_stats_b{B}.txt"), read(f"{G}/{tag}/kernels_by_grid_b{B}.txt")
        if not st:
            continue

        def readable(txt):      # (symbols the trace tool cut before demangling: decode what is there)
            out = []
            for ln in txt.split("\n"):
                m = re.match(r"(_ZN3wis\S+)(\s+)(.*)", ln)
                if m:
                    nm = short(m.group(1))
                    ln = ("wis::" + nm).ljust(len(m.group(1)) + len(m.group(2))) + m.group(3)
                out.append(ln)
            return "\n".join(out)
        st, gr = readable(st), (readable(gr) if gr else gr)
        out = (f"# rocprofv3 --kernel-trace --stats, round {int(RND[1:])}, Whisper large-v2 beam 5, 3.84 s clip, {what}\n\n"
               f"Command (GPU box, `bash tools/gpu_session.sh prof{B}`): `WIS_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --batch {B} --no-cpu-baseline --no-extras`\n"
               "(eager launches: rocprofv3 cannot follow a HIP-graph capture; 7 generate calls + the roofline tap's passes over the decoder weight stream + the one-time weight conversion kernels).\n"
               "Aggregated from the rocpd sqlite output (`kernels` view) by `tools/prof_summary.py`.  Durations of kernels in a dependent chain INCLUDE the boundary before them\n"
               "(start(k+1) = end(k) in the trace), so the column sums to the wall time of the chain.\n\n" + TEMPLATE_NOTE + "\n\n```\n" + st + "```\n")
        if gr:
            out += "\nPer kernel and grid size (threads):\n\n```\n" + gr + "```\n"
        open(f"{P}/{RND}_kernel_stats_large_beam5_b{B}_eager.md", "w").write(out)
        print("wrote kernel stats b", B)


def parse_pmc(path):
    """-> {(kernel name, grid): {counter: (n, mean)}}"""
    out = {}
    txt = read(path)
    if not txt:
        return out
    for line in txt.splitlines():
        m = re.match(r"(.+?)\s+grid=\s*(\d+)\s+(.*)", line)
        if not m:
            continue
        name, grid, rest = m.group(1).strip(), m.group(2), m.group(3)
        cs = {c: (int(n), float(mean)) for c, n, mean in re.findall(r"(\w+): n=(\d+) mean=([\d.eE+-]+)", rest)}
        out.setdefault((name, grid), {}).update(cs)
    return out


def short(name):
    """demangled short form; the trace tools cut long symbols before they demangle them, so integer / bool template arguments of a mangled
    (possibly truncated) name are decoded here: _ZN3wis11gemv_kernelILi1ELi2ELi0ELi1ELb0EEEv... -> gemv_kernel<1, 2, 0, 1, false>"""
    name = name.replace("wis::", "")
    m = re.match(r"_ZN3wis(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base = name[m.end():m.end() + n]
    rest = name[m.end() + n:]
    if rest.startswith("I"):
        args, i = [], 1
        while i < len(rest):
            a = re.match(r"Li(\d+)E|Lb([01])E", rest[i:])
            if a:
                args.append(a.group(1) if a.group(1) is not None else ("true" if a.group(2) == "1" else "false"))
                i += a.end()
                continue
            a = re.match(r"NS_(\d+)", rest[i:])        # a type of namespace wis: NS_<len><name>E
            if a:
                ln = int(a.group(1)); st = i + a.end()
                args.append(rest[st:st + ln]); i = st + ln
                if rest[i:i + 1] == "E":
                    i += 1
                continue
            break
        if args and rest[i:i + 1] == "E":
            return f"{base}<{', '.join(args)}>"
    return base


def pmc_decode(tag):
    res = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE over the eager bench (bash tools/gpu_session.sh pmc fetch FETCH_SIZE <B> / pmc write WRITE_SIZE <B>; "
                   "whisper large-v2 beam 5, steps 2 warmup 1).  Counters are KiB per launch (mean over the launches of that kernel and grid size).  On gfx950 FETCH_SIZE reports HALF of a wide "
                   "(16 B per lane) coalesced read stream, so reads are doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated and small here.  traffic = 2 x FETCH + WRITE; "
                   "algorithmic = the bytes the launch must read once (the weight matrix; for the cross-attention the K and V of the utterances).  The activations of the batched kernels "
                   "(fragment images, 120-480 KiB read by every workgroup) are served by L2 and only partly reach the memory-side counters."}
    for B, ALG in ((1, ALG_B1), (8, ALG_B8)):
        f, w = parse_pmc(f"{G}/{tag}/pmc_fetch_b{B}.txt"), parse_pmc(f"{G}/{tag}/pmc_write_b{B}.txt")
        if not f:
            continue
        rows, tot_t, tot_a = {}, 0.0, 0.0
        for (name, grid), cs in f.items():
            key = f"{short(name)} {grid}"
            if key not in ALG and not short(name).startswith("gemv"):      # (the attention kernels are listed without their template arguments)
                key = f"{short(name).split('<')[0]} {grid}"
            alg = ALG.get(key)
            if alg is None:
                alg = next((v for pat, v in ALG.items() if re.fullmatch(pat, key)), None)
            if alg is None or "FETCH_SIZE" not in cs:
                continue
            n, fetch = cs["FETCH_SIZE"]
            wr = w.get((name, grid), {}).get("WRITE_SIZE", (0, 0.0))[1]
            traffic = 2.0 * fetch * 1024 + wr * 1024
            rows[key] = {"launches": n, "fetch_KiB_mean": round(fetch, 1), "write_KiB_mean": round(wr, 1), "traffic_bytes_per_launch": round(traffic), "algorithmic_bytes_per_launch": alg,
                         "traffic_over_algorithmic": round(traffic / alg, 3)}
            if "gemv" in key:
                tot_t += traffic * n; tot_a += alg * n
        res[f"batch_{B}"] = {"per_kernel": rows, "skinny_gemm_traffic_over_algorithmic": round(tot_t / tot_a, 3) if tot_a else None}
    b1 = res.get("batch_1", {}).get("skinny_gemm_traffic_over_algorithmic")
    if b1:
        res["kernel"] = "wis::gemv_kernel (decoder skinny GEMM, one utterance)"
        res["algorithmic_bytes_per_launch"] = 8294294
        res["traffic_over_algorithmic"] = b1
        res["hbm_bytes_per_launch"] = round(b1 * 8294294)
    json.dump(res, open(f"{P}/{RND}_pmc_decode.json", "w"), indent=1)
    print(f"wrote {RND}_pmc_decode.json", b1, res.get("batch_8", {}).get("skinny_gemm_traffic_over_algorithmic"))


def pmc_encoder(tag, tcc_tag=None):
    a, b = parse_pmc(f"{G}/{tag}/pmc_sq1_b8.txt"), parse_pmc(f"{G}/{tag}/pmc_sq2_b8.txt")
    if not a:
        return
    lines = [f"# SQ counters of the encoder kernels at 8 utterances per device batch (rocprofv3 --pmc, two passes), round {int(RND[1:])}", "",
             "Command per pass (`bash tools/gpu_session.sh pmc sq1 \"...\" 8`): `WIS_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc <8 SQ counters> --output-format csv -- python bench.py --steps 2 --warmup 1 --batch 8 "
             "--no-cpu-baseline --no-extras --no-roofline` (large-v2).  `SQ_WAVE_CYCLES`, `SQ_WAIT_*`, `SQ_ACTIVE_INST_*` count quad-cycles; `SQ_VALU_MFMA_BUSY_CYCLES` counts cycles "
             "(16 per `v_mfma_f32_16x16x32_f16`, 32 per `32x32x16`), so the MFMA share of wave time = MFMA_BUSY / (4 x WAVE_CYCLES) - with two waves per SIMD (the 8-phase GEMM) the matrix pipe of a SIMD is busy "
             "for twice that share.  Means per launch.  (r4: the attention row is `enc_attn_lazy_kernel`; round 3's `enc_attn_kernel` row read 0.27 / 0.39 (0.05) / 0.34 / 0.15, vector ALU 0.26 of wave time = 0.78 of a SIMD, 4612 VALU instructions per wave.)", "",
             "| kernel (grid threads) | launches | parked on waitcnt / barrier (WAIT_ANY) | issue stall (WAIT_INST_ANY; of which LDS) | issuing (ACTIVE_INST_ANY) | MFMA busy / wave time | MFMA pipe busy per SIMD (x waves per SIMD) | LDS bank-conflict / LDS active cycles | vector ALU busy / wave time (ACTIVE_INST_VALU; x3 = share of a SIMD at three waves) | VALU instructions per wave (INSTS_VALU / waves) |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for (name, grid), cs in sorted(a.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", (0, 0))[1] * kv[1].get("SQ_WAVE_CYCLES", (0, 0))[0]):
        if not any(k in name for k in ("gemm_8p", "gemm_f16", "gemm_pp", "enc_attn", "layernorm")):
            continue
        wc = cs.get("SQ_WAVE_CYCLES", (0, 0))[1]
        if wc <= 0:
            continue
        g = lambda c: cs.get(c, (0, 0.0))[1]
        c2 = b.get((name, grid), {})
        waves = 2 if "gemm_8p" in name or "gemm_pp" in name else 1
        mfma = g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * wc)
        lines.append(f"| `{short(name)}` ({grid}) | {cs['SQ_WAVE_CYCLES'][0]} | {g('SQ_WAIT_ANY') / wc:.2f} | {g('SQ_WAIT_INST_ANY') / wc:.2f} ({g('SQ_WAIT_INST_LDS') / wc:.2f}) | {g('SQ_ACTIVE_INST_ANY') / wc:.2f} | "
                     f"{mfma:.2f} | {mfma * waves:.2f} | {c2.get('SQ_LDS_BANK_CONFLICT', (0, 0))[1]:.3g} / {c2.get('SQ_LDS_IDX_ACTIVE', (0, 0))[1]:.3g} | "
                     f"{c2.get('SQ_ACTIVE_INST_VALU', (0, 0))[1] / wc:.2f} | {c2.get('SQ_INSTS_VALU', (0, 0))[1] / (int(grid) / 64):.0f} |")
    t = parse_pmc(f"{G}/{tcc_tag}/pmc_tcc_b1.txt") if tcc_tag else {}
    if t:
        lines += ["", "## L2 (TCC) counters of the encoder kernels at ONE utterance (`pmc tcc \"TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum\" 1`)", "",
                  "Requests are 128-byte lines.  The one-utterance GEMMs pull ~270 MB through the L2s per launch for ~17 MB of unique operands; a third of the requests miss "
                  "(every XCD fetches its own copy of the shared panels, and the workgroups of a panel ask for the same k-slice at the same time).  Taken on the intermediate build that ran QKV, FFN1 and the FFN2 slices on the 128 x 256 8-phase tile (the product keeps it for FFN1).", "",
                  "| kernel (grid threads) | launches | L2 requests | hits | misses | miss share | reads from the fabric (EA) |", "|---|---|---|---|---|---|---|"]
        for (name, grid), cs in sorted(t.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", (0, 0))[1] * kv[1].get("TCC_REQ_sum", (0, 0))[0]):
            if not any(k in name for k in ("gemm_8p", "gemm_f16", "gemm_pp", "enc_attn")):
                continue
            g = lambda c: cs.get(c, (0, 0.0))[1]
            if g("TCC_REQ_sum") <= 0:
                continue
            lines.append(f"| `{short(name)}` ({grid}) | {cs['TCC_REQ_sum'][0]} | {g('TCC_REQ_sum'):.3g} | {g('TCC_HIT_sum'):.3g} | {g('TCC_MISS_sum'):.3g} | {g('TCC_MISS_sum') / g('TCC_REQ_sum'):.2f} | {g('TCC_EA0_RDREQ_sum'):.3g} |")
    open(f"{P}/{RND}_pmc_encoder_sq.md", "w").write("\n".join(lines) + "\n")
    print(f"wrote {RND}_pmc_encoder_sq.md")


def lab(tag_multi, tag_single):
    out = ["# tools/gemm_lab on MI355X, round 3: the encoder GEMM loop variants on the four per-layer shapes of large-v2", "",
           "`tools/bin/gemm_lab <M> 0 1` (weights rotated through 640 MB: W streams from HBM as in the encoder; 40 timed launches per variant; every variant checked against the 128 x 128 product loop).",
           "Variants: `v0 / v1` register-staged loops (late / early staging order), `g2 / g3` LDS-DMA with 2 / 3 buffers, `pp` ping-pong groups, **`8p`** the 8-phase 256 x 256 LDS-DMA kernel "
           "(csrc/enc_kernels.hip gemm_8p_kernel), `8p-nostagger` without the one-barrier offset between the wave rows, `8p-noprio` without s_setprio around the MFMA clusters, "
           "`8p-nostore` (ablation, wrong results) without the epilogue stores, `8p-ldsep` epilogue through LDS with 16-byte full-line stores, `8p-persist*` one workgroup per CU looping over tiles.", ""]
    for tag, what in ((tag_multi, "8 utterances (M = 12000)"), (tag_single, "one utterance (M = 1500) and 8 utterances, first take of the 8-phase kernel with its epilogue ablations")):
        d = f"{G}/{tag}"
        if not os.path.isdir(d):
            continue
        for fn in sorted(os.listdir(d)):
            if fn.startswith("lab_M"):
                txt = "".join(l for l in open(f"{d}/{fn}") if not l.startswith("  stamps") and "steady" not in l)
                out += [f"## gpurun_out/{tag}/{fn} - {what}", "", "```", txt.rstrip(), "```", ""]
    open(f"{P}/{RND}_gemm_lab.md", "w").write("\n".join(out) + "\n")
    print("wrote r03_gemm_lab.md")


def bench(tag):
    src = f"{G}/{tag}/bench_default.json"
    if os.path.exists(src):
        line = [l for l in open(src) if l.startswith("{")][-1]
        json.dump(json.loads(line), open(f"{P}/{RND}_bench_large_beam5.json", "w"), indent=1)
        print(f"wrote {RND}_bench_large_beam5.json")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r03")
    ap.add_argument("--stats"); ap.add_argument("--pmc"); ap.add_argument("--lab"); ap.add_argument("--lab-single"); ap.add_argument("--bench"); ap.add_argument("--tcc")
    ap.add_argument("--pmc-enc", help="only the encoder SQ table (passes sq1 / sq2 of that tag)")
    a = ap.parse_args()
    globals()["RND"] = a.round
    if a.stats: kernel_stats(a.stats)
    if a.pmc: pmc_decode(a.pmc); pmc_encoder(a.pmc, a.tcc)
    if a.pmc_enc: pmc_encoder(a.pmc_enc, a.tcc)
    if a.lab or a.lab_single: lab(a.lab, a.lab_single)
    if a.bench: bench(a.bench)
