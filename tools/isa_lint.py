"""Build-time lint over the gfx950 code objects of the library (willow-inference-server_amd/build.py runs it after every build).

    python tools/isa_lint.py [--table] build/*.hip.o

Rule: NO kernel that contains MFMA instructions may contain packed-f32 VALU arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32),
(rule 2, round 6: no instruction names the destination SGPR of a scalar load before the s_waitcnt that retires it - smem_hazards below;
rule 3: no AGPRs in the encoder attention / GEMM kernels = `-amdgpu-mfma-vgpr-form` in effect; rule 4: the decoder kernels' descriptors preload kernel
arguments = `-amdgpu-kernarg-preload-count` in effect),
whatever its occupancy (the failures were seen at three waves per SIMD, <= 168 unified VGPRs; the table prints the register bound
so a reader can see which kernels could get there), and no kernel may use scratch.  Round 3/4 finding ("The two-tile
kernel's corruption"): the two-n-tile skinny GEMM's 168-VGPR instantiation returned wrong LOW halves of v_pk_*_f32 results in
lanes 48-63 in about half of its launches on every box it was tried on with hipcc's SLP-vectorised epilogue; the same source with
scalar f32 arithmetic (same registers, same occupancy) and every two-waves-per-SIMD form are clean.  The static scan
(tools/isa_hazard_scan.py) shows no MFMA-result -> packed-VALU register dependence in that kernel (the accumulators pass through
LDS and a barrier first), so no ISA wait-state rule explains it; until it is understood the combination is kept out of the
library by this check instead of by convention.

The occupancy bound is the register bound only (LDS and workgroup size can only lower it): conservative."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("WIS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
PK = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")
MAX_VGPR_3_WAVES = 168          # 512 unified VGPRs per SIMD lane, allocation granule 8: 3 x 168 = 504


def code_object(path, tmp):
    """device code object of a host object / shared library built by hipcc (None if it carries no gfx950 image)"""
    fb = os.path.join(tmp, os.path.basename(path) + ".fatbin")
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, path, os.devnull], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fb) or os.path.getsize(fb) == 0:
        return None
    co = fb + ".co"
    r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fb, "--targets=" + TARGET, "--output=" + co], capture_output=True, text=True)
    return co if r.returncode == 0 and os.path.exists(co) else None


def kernels(co):
    """-> {kernel: dict(vgpr, agpr, mfma, pk, n_instr)}"""
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    meta, cur = {}, {}
    for ln in notes.split("\n"):
        m = re.match(r"\s*-?\s*\.(agpr_count|vgpr_count|name|private_segment_fixed_size):\s*(\S+)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if ln.lstrip().startswith("- ."):          # first key of a kernel record
            cur = {}
        cur[k] = v
        if "name" in cur and "vgpr_count" in cur:
            meta[cur["name"]] = {"vgpr": int(cur["vgpr_count"]), "agpr": int(cur.get("agpr_count", 0)), "scratch": int(cur.get("private_segment_fixed_size", 0))}
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    out, name = {}, None
    for ln in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            name = m.group(1)
            if name in meta:
                out[name] = dict(meta[name], mfma=0, pk=0, n_instr=0)
            else:
                name = None
            continue
        if name is None:
            continue
        t = ln.strip()
        if not t:
            continue
        out[name]["n_instr"] += 1
        if t.startswith("v_mfma") or t.startswith("v_smfmac"):
            out[name]["mfma"] += 1
        elif PK.search(t):
            out[name]["pk"] += 1
    return out


SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def _sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def smem_hazards(co):
    """Round 6 (advisor, round 5): csrc/common.hpp uniform_load_issue_* requests workgroup-uniform operands with `s_load_dword` from inline asm and
    the caller waits by hand (uniform_load_wait) - the compiler believes the destination SGPR is defined at the asm statement, so nothing but the
    helper's own discipline keeps a copy / spill / use of that register from being scheduled between the request and the wait.  Rule, checked on
    the generated code of EVERY scalar load (the compiler's own obey it by construction): between an `s_load_*` and the next `s_waitcnt` that
    retires it (lgkmcnt(0): scalar loads return out of order) no instruction may name a destination SGPR of the load.  Checked inside basic
    blocks (the pending set is dropped at branches and at branch targets, where another path joins).  -> [(kernel, load line, offending line)]"""
    return smem_hazards_text(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout)


def smem_hazards_text(dis):
    """the rule of smem_hazards over an `llvm-objdump -d --no-show-raw-insn` listing"""
    # per kernel: [(address, instruction text)] and the set of branch-target addresses (basic-block starts)
    kernels_, cur, base = {}, None, 0
    for ln in dis.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", ln)
        if m:
            cur, base = m.group(2), int(m.group(1), 16)
            kernels_[cur] = ([], set())
            continue
        if cur is None or "//" not in ln:
            continue
        t, _, c = ln.partition("//")
        t = t.strip()
        ma = re.match(r"\s*([0-9A-Fa-f]+):", c)
        if not t or not ma:
            continue
        kernels_[cur][0].append((int(ma.group(1), 16), t))
        if t.startswith("s_cbranch") or t.startswith("s_branch"):
            mt = re.search(r"<.*\+0x([0-9a-fA-F]+)>\s*$", c)
            kernels_[cur][1].add(base + int(mt.group(1), 16) if mt else -1)
    out = []
    for name, (ins, targets) in kernels_.items():
        pending = {}
        for addr, t in ins:
            if addr in targets:          # another path joins here: what is pending on it is not known (the compiler's own loads are waited per path)
                pending = {}
            op = t.split()[0]
            if op == "s_waitcnt":
                if "lgkmcnt(0)" in t or re.fullmatch(r"s_waitcnt\s+(0x)?[0-9a-f]+", t):
                    pending = {}
                continue
            if op.startswith("s_cbranch") or op.startswith("s_branch") or op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                pending = {}
                continue
            used = _sregs(t[len(op):])
            is_load = op.startswith("s_load") or op.startswith("s_buffer_load")
            if pending:
                ops = t[len(op):].split(",")
                hit = (_sregs(",".join(ops[1:])) if is_load else used) & set(pending)      # (a load may not build its address from a register still in flight)
                if hit:
                    out.append((name, pending[min(hit)], t))
            if is_load:
                for r in _sregs(t[len(op):].split(",")[0]):
                    pending[r] = t
    return out


PRELOAD_KERNELS = ("gemv_kernel", "gemv_dual_kernel", "gemv_frag_kernel", "gemv_frag2_kernel", "gemv_frag3_kernel", "gemv_frag_ms_kernel", "dec_self_attn_kernel", "dec_cross_attn_kernel", "dec_cross_attn_rs_kernel")


def kernarg_preload(co):
    """-> {kernel: dwords of kernel arguments the command processor preloads into SGPRs} from the kernel descriptors (`<kernel>.kd`, 64 bytes in .rodata:
    bits 0-6 of the 16-bit field at offset 58).  Rule 4 (round 6): the library is built with `-mllvm -amdgpu-kernarg-preload-count=16` and the decoder
    kernels list their early operands as leading scalars so that the switch takes effect (decode step -4.2 %); a compiler that drops
    the switch would leave them correct and slower - so a decoder kernel with a preload length of 0 fails the build."""
    syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co], capture_output=True, text=True, check=True).stdout
    secs = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", "-W", co], capture_output=True, text=True, check=True).stdout
    m = re.search(r"\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
    if not m:
        return {}
    addr, off, size = (int(x, 16) for x in m.groups())
    blob = open(co, "rb").read()[off:off + size]
    out = {}
    for ln in syms.split("\n"):
        f = ln.split()
        if len(f) >= 8 and f[-1].endswith(".kd") and f[3] == "OBJECT":
            a = int(f[1], 16) - addr
            if 0 <= a and a + 64 <= len(blob):
                out[f[-1][:-3]] = int.from_bytes(blob[a + 58:a + 60], "little") & 0x7F
    return out


def lint(paths, table=False):
    bad, rows = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for p in paths:
            co = code_object(p, tmp)
            if co is None:
                continue
            for kn, ld, use in smem_hazards(co):
                bad.append((os.path.basename(p), f"{kn}: `{use}` names an SGPR of the scalar load `{ld}` before the wait that retires it", 0, 0, 0, -1, 0))
            for kn, n_pre in sorted(kernarg_preload(co).items()):
                if n_pre == 0 and any(f"wis{len(t)}{t}" in kn or f"wis::{t}" in kn for t in PRELOAD_KERNELS):
                    bad.append((os.path.basename(p), f"{kn}: no kernel argument is preloaded - `-mllvm -amdgpu-kernarg-preload-count` is not in effect (or the kernel's signature no longer leads with scalars)", 0, 0, 0, -1, 0))
            for name, k in sorted(kernels(co).items()):
                waves = min(8, 512 // (((k["vgpr"] + 7) // 8) * 8)) if k["vgpr"] else 8
                rows.append((os.path.basename(p), name, k["vgpr"], waves, k["mfma"], k["pk"], k["scratch"]))
                if k["mfma"] and k["pk"]:
                    bad.append(rows[-1])
                # rule 3 (round 6; round-5 review weak 11: "one ROCm bump from a silent change"): the library is built with `-mllvm -amdgpu-mfma-vgpr-form`
                # (accumulators in VGPRs: no v_accvgpr moves around the softmax of the attention loops); if a compiler stops honouring the switch the
                # kernels still run, 10-15 % slower - so an MFMA kernel that allocates AGPRs fails the build instead
                # (checked on the kernels the switch was introduced for - the encoder attention loops and GEMMs; a register-starved skinny-GEMM
                # instantiation may legitimately use AGPRs as spill space)
                if k["mfma"] and k["agpr"] and ("enc_attn" in name or "gemm_" in name):
                    bad.append((os.path.basename(p), f"{name}: {k['mfma']} MFMAs with {k['agpr']} AGPRs allocated - `-mllvm -amdgpu-mfma-vgpr-form` is not in effect", 0, 0, 0, -1, 0))
    if table:
        print(f"{'object':22s} {'VGPRs':>5s} {'waves/SIMD':>10s} {'MFMA':>6s} {'v_pk f32':>8s}  kernel")
        for o, n, v, w, mf, pk, sc in rows:
            if mf or pk or sc:
                print(f"{o:22s} {v:5d} {w:10d} {mf:6d} {pk:8d}  {n}" + (f"  [scratch {sc} B]" if sc else ""))
    return bad, rows


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    bad, rows = lint(args, table="--table" in sys.argv)
    for o, n, v, w, mf, pk, sc in bad:
        if pk < 0:
            print(f"isa_lint: {o}: {n}", file=sys.stderr)
            continue
        print(f"isa_lint: {o}: {n}: {mf} MFMAs and {pk} packed-f32 VALU instructions at {v} VGPRs ({w} waves per SIMD possible)", file=sys.stderr)
    scratch = [r for r in rows if r[6]]
    for o, n, v, w, mf, pk, sc in scratch:
        print(f"isa_lint: {o}: {n}: {sc} bytes of scratch per lane", file=sys.stderr)
    print(f"isa_lint: {len(rows)} kernels, {sum(1 for b in bad if b[5] >= 0)} packed-f32 violations, {sum(1 for b in bad if b[5] < 0)} scalar-load hazards / AGPR findings, {len(scratch)} kernels with scratch")
    sys.exit(1 if bad or scratch else 0)
