"""Build libwis_hip.so (gfx950) in-tree: hipcc for the HIP sources, gcc for the plain-C audio IO.

    python willow-inference-server_amd/build.py [--force]

The .so lands in willow-inference-server_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libwis_hip.so")

HIP_SOURCES = ["logmel.hip", "enc_kernels.hip", "dec_kernels.hip", "model.hip"]
C_SOURCES = ["audio_io.c"]
HEADERS = ["common.hpp", "kernels.hpp", os.path.join(ROOT, "include", "wis_hip.h")]
EXPORTS = os.path.join(CSRC, "exports.map")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA accumulators live in VGPRs (gfx90a+ unified register file) instead of AGPRs.  hipcc's default put
# the encoder attention's score tiles in AGPRs and moved them with 64 v_accvgpr_write / v_accvgpr_read per key tile and wave
# (zero-initialisation and hand-over to the softmax arithmetic, ~20 % of the loop's vector-ALU time); in VGPR form the first MFMA
# of a tile takes the inline constant 0 as its C operand and the softmax reads the result registers directly.  No kernel spills.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
             "-Wno-unused-result", "-Wno-unused-value", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize",
             "-mllvm", "-amdgpu-kernarg-preload-count=16"] + os.environ.get("WIS_EXTRA_HIPFLAGS", "").split()
C_FLAGS = ["-O2", "-fPIC", "-std=c11", "-I" + os.path.join(ROOT, "include")]
# -amdgpu-kernarg-preload-count=16 (r5): the command processor hands the first kernel arguments to a wave in SGPRs at launch instead of
# the wave fetching them from the kernarg segment - one HBM round trip (the decode step's 1.6 GB weight stream leaves nothing cached between
# graph replays) in front of the first address computation of EVERY one of the step's 229 kernels.  hipcc preloads only leading scalar /
# pointer arguments (a by-value struct gets kernarg_preload_length 0 - which is why round 1 measured "no change" for this flag: the skinny
# GEMMs took one struct), so the decoder kernels now list what their first round of loads needs as leading scalars, at most 14 dwords
# (csrc/dec_kernels.hip WIS_GV_LEAD, dec_self_attn_kernel, dec_cross_attn_kernel).  Same box, same call: decode step 1.402 -> 1.342 ms at
# one utterance (-4.2 %), 2.215 -> 2.146 ms at eight (-3.1 %), whole utterance 29.47 -> 28.42 ms.
# -fno-slp-vectorize, every HIP source: hipcc's SLP pass turns adjacent scalar f32 adds / multiplies (epilogue arithmetic on float4
# values) into packed-f32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), and the two-n-tile skinny GEMM's
# 3-waves-per-SIMD instantiation (168 VGPRs) returned WRONG low halves of those packed results in lanes 48-63 in about half of its
# launches on every box it was tried on (round 3's "features 12 and 14" corruption: 10 808 of 96 000 launches wrong with SLP, 0 of
# 240 000 without, same registers, same occupancy, same chip - tools/frag_stress.hip, tools/frag2_lab.hip).
# Scalar f32 code is also what the guide recommends beside MFMAs (packed f32 is "an anti-lever" there).  The flag alone does not
# remove every packed op (sums on ext-vector types still lower to v_pk_add_f32: the encoder epilogues add component-wise, add4), so
# the rule is ENFORCED on the built code objects: tools/isa_lint.py, run by build() below - no kernel with MFMAs may contain
# v_pk_{add,mul,fma}_f32 (today no kernel of the library contains one at all), and no kernel may use scratch (a spill inside the 8-phase GEMM loop
# would move its hand-counted vmcnt waits).
SOURCE_FLAGS = {}


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(HIP_FLAGS + C_FLAGS + [f"{k}:{' '.join(v)}" for k, v in sorted(SOURCE_FLAGS.items())]).encode())
    return h.hexdigest()


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build failed: " + os.path.basename(cmd[-1]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def build(force=False, verbose=True, variant=None, extra_flags=()):
    """variant: tuning builds next to the product library (lib/libwis_hip_<variant>.so, selected at run time with
    WIS_LIB_PATH; objects under build/<variant>/) compiled with `extra_flags` on top of the product flags."""
    objdir = os.path.join(OBJDIR, variant) if variant else OBJDIR
    lib = os.path.join(LIBDIR, f"libwis_hip_{variant}.so") if variant else LIB
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    hip_flags = HIP_FLAGS + list(extra_flags)
    for src in HIP_SOURCES + C_SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        stamp = obj + ".sha"
        dig = _digest([path] + hdrs) + "|" + " ".join(extra_flags)
        if force or not os.path.exists(obj) or not os.path.exists(stamp) or open(stamp).read() != dig:
            if verbose:
                print("[build] compiling", src, flush=True)
            if src.endswith(".c"):
                _run(["gcc"] + C_FLAGS + ["-c", path, "-o", obj])
            else:
                _run([HIPCC] + hip_flags + SOURCE_FLAGS.get(src, []) + ["-c", path, "-o", obj])
            with open(stamp, "w") as f:
                f.write(dig)
        objs.append(obj)
    newest = max(os.path.getmtime(o) for o in objs)
    newest = max(newest, os.path.getmtime(EXPORTS))
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < newest:
        if verbose:
            print("[build] linking", os.path.relpath(lib, ROOT), flush=True)
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + EXPORTS] + objs + ["-o", lib])
    if not variant or os.environ.get("WIS_LINT_VARIANTS"):
        lint_objects([o for o in objs if o.endswith(".hip.o")])
    return lib


def lint_objects(objs):
    """tools/isa_lint.py over the device code objects: fails the build on a packed-f32 / MFMA / >= 3 waves per SIMD kernel or on scratch use"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wis_isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, rows = mod.lint(objs)
    scratch = [r for r in rows if r[6]]
    if bad or scratch:
        for o, n, v, w, mf, pk, sc in bad:
            if pk < 0:
                sys.stderr.write(f"[build] isa_lint: {o}: {n}\n")
                continue
            sys.stderr.write(f"[build] isa_lint: {o}: {n}: {mf} MFMAs + {pk} packed-f32 VALU instructions at {v} VGPRs ({w} waves per SIMD possible)\n")
        for o, n, v, w, mf, pk, sc in scratch:
            sys.stderr.write(f"[build] isa_lint: {o}: {n}: {sc} bytes of scratch per lane (register spill)\n")
        raise RuntimeError(f"build failed: isa_lint ({len(bad)} packed-f32 violations, {len(scratch)} kernels with scratch)")
    return len(rows)


if __name__ == "__main__":
    # python build.py [--force] [--variant NAME -DFLAG -fflag ...]
    argv = sys.argv[1:]
    var = argv[argv.index("--variant") + 1] if "--variant" in argv else None
    print(build(force="--force" in argv, variant=var, extra_flags=[a for a in argv if a.startswith("-") and a not in ("--force", "--variant")]))
