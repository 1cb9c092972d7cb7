"""Whisper weight sets for the wis_hip engine.

The reference loads CTranslate2 model directories (`models/tovera-wis-whisper-*`, reference
main.py:341-444, utils.sh:99-108).  No checkpoint exists offline, so this module provides
  * the architecture table of the five sizes WIS serves (main.py:564-573),
  * seeded synthetic weights at the TRUE shapes (N(0, 0.02^2) linears/convs/embeddings,
    LayerNorm gamma 1 / beta 0) under the CTranslate2 WhisperSpec variable names,
  * the flat arena + tensor index that `wis_model_create` consumes,
  * a reader for a real CTranslate2 `model.bin` (format: SURVEY Appendix C) when one is mounted.
The oracle (oracle/whisper_ref.py) consumes the same name -> array dict, so both sides see
bit-identical (f16-rounded) weights.
"""
import json
import os
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

# size -> (d_model, layers (enc = dec), heads); "large" is large-v2 (utils.sh:264-271)
ARCH = {
    "tiny": (384, 4, 6),
    "base": (512, 6, 8),
    "small": (768, 12, 12),
    "medium": (1024, 24, 16),
    "large": (1280, 32, 20),
}
N_VOCAB = 51865
N_AUDIO_CTX = 1500
N_TEXT_CTX = 448
N_MELS = 80

# special ids of the multilingual vocabulary
EOT, SOT, TRANSLATE, TRANSCRIBE, NO_SPEECH, NO_TIMESTAMPS = 50257, 50258, 50358, 50359, 50362, 50363
LANG_IDS = list(range(50259, 50358))
# CT2 config.json:suppress_ids of the released multilingual checkpoints = the 86 non-speech ids
# of the HF generation config + <|translate|> + <|transcribe|>
SUPPRESS_IDS = [1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 359, 503, 522, 542, 873,
                893, 902, 918, 922, 931, 1350, 1853, 1982, 2460, 2627, 3246, 3253, 3268, 3536, 3846, 3961, 4183, 4667, 6585,
                6647, 7273, 9061, 9383, 10428, 10929, 11938, 12033, 12331, 12562, 13793, 14157, 14635, 15265, 15618, 16553,
                16604, 18362, 18956, 20075, 21675, 22520, 26130, 26161, 26435, 28279, 29464, 31650, 32302, 32470, 36865,
                42863, 47425, 49870, 50254, 50258, 50358, 50359, 50360, 50361, 50362]
SUPPRESS_IDS_BEGIN = [220, EOT]


def arch(size):
    if size == "large-v2":
        size = "large"
    d, L, H = ARCH[size]
    return dict(size=size, d_model=d, n_layers=L, n_heads=H, n_vocab=N_VOCAB, n_audio_ctx=N_AUDIO_CTX, n_text_ctx=N_TEXT_CTX, n_mels=N_MELS)


def tensor_shapes(d, L, n_vocab=N_VOCAB, n_text_ctx=N_TEXT_CTX):
    """name -> (shape, kind) in CTranslate2 WhisperSpec naming; kind in {w, b, g, beta, emb}."""
    t = {}
    t["encoder/conv1/weight"] = ((d, N_MELS, 3), "w")
    t["encoder/conv1/bias"] = ((d,), "b")
    t["encoder/conv2/weight"] = ((d, d, 3), "w")
    t["encoder/conv2/bias"] = ((d,), "b")
    t["encoder/position_encodings/encodings"] = ((N_AUDIO_CTX, d), "pos")     # fixed sinusoids, float32
    for side, n in (("encoder", L), ("decoder", L)):
        for l in range(n):
            p = f"{side}/layer_{l}/"
            t[p + "self_attention/layer_norm/gamma"] = ((d,), "g")
            t[p + "self_attention/layer_norm/beta"] = ((d,), "beta")
            t[p + "self_attention/linear_0/weight"] = ((3 * d, d), "w")
            t[p + "self_attention/linear_0/bias"] = ((3 * d,), "bqkv")
            t[p + "self_attention/linear_1/weight"] = ((d, d), "w")
            t[p + "self_attention/linear_1/bias"] = ((d,), "b")
            if side == "decoder":
                t[p + "attention/layer_norm/gamma"] = ((d,), "g")
                t[p + "attention/layer_norm/beta"] = ((d,), "beta")
                t[p + "attention/linear_0/weight"] = ((d, d), "w")
                t[p + "attention/linear_0/bias"] = ((d,), "b")
                t[p + "attention/linear_1/weight"] = ((2 * d, d), "w")
                t[p + "attention/linear_1/bias"] = ((2 * d,), "bkv")
                t[p + "attention/linear_2/weight"] = ((d, d), "w")
                t[p + "attention/linear_2/bias"] = ((d,), "b")
            t[p + "ffn/layer_norm/gamma"] = ((d,), "g")
            t[p + "ffn/layer_norm/beta"] = ((d,), "beta")
            t[p + "ffn/linear_0/weight"] = ((4 * d, d), "w")
            t[p + "ffn/linear_0/bias"] = ((4 * d,), "b")
            t[p + "ffn/linear_1/weight"] = ((d, 4 * d), "w")
            t[p + "ffn/linear_1/bias"] = ((d,), "b")
        t[f"{side}/layer_norm/gamma"] = ((d,), "g")
        t[f"{side}/layer_norm/beta"] = ((d,), "beta")
    t["decoder/embeddings/weight"] = ((n_vocab, d), "emb")
    t["decoder/position_encodings/encodings"] = ((n_text_ctx, d), "w")
    return t


def sinusoids(length=N_AUDIO_CTX, channels=1280):
    """Whisper encoder positions, float32 like the HF / openai tables."""
    inc = np.log(10000.0) / (channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2)).astype(np.float32)
    t = np.arange(length, dtype=np.float32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def synthetic_weights(size, seed=1234, std=0.02, emb_std=None, ln_jitter=0.0, threads=8, n_vocab=N_VOCAB):
    """Deterministic synthetic weights (f16) for `size`; each tensor has its own stream so generation order and
    threading do not matter.  K-projection bias is zero (Whisper's k_proj has no bias)."""
    a = arch(size)
    d, L = a["d_model"], a["n_layers"]
    shapes = tensor_shapes(d, L, n_vocab)
    emb_std = std if emb_std is None else emb_std

    def gen(item):
        name, (shape, kind) = item
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        if kind == "pos":
            return name, sinusoids(shape[0], shape[1])
        if kind == "g":
            v = np.ones(shape, np.float32)
            if ln_jitter:
                v += ln_jitter * rng.standard_normal(shape, dtype=np.float32)
        elif kind == "beta":
            v = np.zeros(shape, np.float32)
            if ln_jitter:
                v += ln_jitter * rng.standard_normal(shape, dtype=np.float32)
        else:
            v = rng.standard_normal(shape, dtype=np.float32)
            v *= np.float32(emb_std if kind == "emb" else std)
            if kind == "bqkv":
                v[d:2 * d] = 0.0
            elif kind == "bkv":
                v[:d] = 0.0
        return name, v.astype(np.float16)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        out = dict(ex.map(gen, shapes.items()))
    return out


def arena_layout(shapes_dtypes):
    """[(name, shape, numpy dtype)] -> (index list of dicts, total bytes).  Depends on shapes only, so every rank of a
    multi-GPU job can compute it without holding the data (the arena itself arrives by RCCL broadcast)."""
    index, off = [], 0
    for name, shape, dt in shapes_dtypes:
        dt = np.dtype(dt)
        assert dt in (np.dtype(np.float16), np.dtype(np.float32)), (name, dt)
        off = (off + 255) & ~255
        index.append(dict(name=name, dtype="f16" if dt == np.float16 else "f32", shape=[int(x) for x in shape], offset=off))
        off += int(np.prod(shape)) * dt.itemsize
    return index, off


def synthetic_layout(size, n_vocab=N_VOCAB):
    a = arch(size)
    shapes = tensor_shapes(a["d_model"], a["n_layers"], n_vocab)
    return arena_layout([(n, s, np.float32 if k == "pos" else np.float16) for n, (s, k) in shapes.items()])


def build_arena(weights):
    """name -> ndarray (f16 / f32)  =>  (arena uint8 ndarray, index list of dicts)."""
    index, total = arena_layout([(n, v.shape, v.dtype) for n, v in weights.items()])
    arena = np.zeros(total, np.uint8)
    for e in index:
        v = np.ascontiguousarray(weights[e["name"]])
        arena[e["offset"]:e["offset"] + v.nbytes] = v.view(np.uint8).reshape(-1)
    return arena, index


def read_ct2_model_bin(path, return_attrs=False):
    """Reader for a CTranslate2 `model.bin` (binary version 6, WhisperSpec; layout per SURVEY Appendix C — written from
    recall of CTranslate2 4.1.0, unverified offline).  Returns name -> float ndarray (and, with return_attrs, the integer
    attribute variables).

    A real WhisperSpec file also carries non-tensor variables: rank-0 scalars such as `encoder/num_heads`,
    `decoder/num_heads` (int16), `decoder/alignment_layer` / `alignment_heads` (int16), `decoder/activation` (int8) and int8
    flags (`pre_norm`, `scale_embeddings`, ...).  They are attributes, not weights: they are collected separately and never
    reach the arena.  What IS rejected is a quantised weight MATRIX (int8 / int16 of rank >= 1 with a companion
    `weight_scale`): WIS's models are exported with --quantization float16 (utils.sh:71,104); this engine quantises the
    decoder itself for compute_type int8_float16."""
    dtypes = {0: np.float32, 1: np.int8, 2: np.int16, 3: np.int32, 4: np.float16, 5: "bfloat16"}
    out, attrs, aliases = {}, {}, {}
    with open(path, "rb") as f:
        def rd(fmt):
            return struct.unpack("<" + fmt, f.read(struct.calcsize("<" + fmt)))

        def rstr():
            (n,) = rd("H")
            return f.read(n)[:-1].decode()

        (ver,) = rd("I")
        spec = rstr()
        (rev,) = rd("I")
        if "Whisper" not in spec:
            raise ValueError(f"{path}: spec {spec!r} is not a Whisper model")
        (nvar,) = rd("I")
        scaled = set()
        for _ in range(nvar):
            name = rstr()
            (rank,) = rd("B")
            dims = rd("I" * rank)
            (dt,) = rd("B")
            (nb,) = rd("I")
            raw = f.read(nb)
            if dt not in dtypes:
                raise ValueError(f"{name}: unknown dtype id {dt}")
            if dtypes[dt] == "bfloat16":
                v = (np.frombuffer(raw, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(dims).copy()
            else:
                v = np.frombuffer(raw, dtype=dtypes[dt]).reshape(dims).copy()
            if name.endswith("weight_scale"):
                scaled.add(name[:-len("_scale")])
                continue
            if v.dtype.kind == "i":
                if v.size <= 64 and not name.endswith("/weight"):          # scalar / small integer attribute
                    attrs[name] = v
                    continue
                raise ValueError(f"{name}: int{8 * v.dtype.itemsize} weight matrix {tuple(dims)} - quantised exports are not supported "
                                 "(re-export with --quantization float16, utils.sh:104)")
            out[name] = v
        for n in scaled:
            if n in out and out[n].dtype.kind != "f":
                raise ValueError(f"{n}: quantised export")
        (nal,) = rd("I")
        for _ in range(nal):
            al = rstr()
            aliases[al] = rstr()
    for al, tgt in aliases.items():
        if tgt in out:
            out.setdefault(al, out[tgt])
        elif tgt in attrs:
            attrs.setdefault(al, attrs[tgt])
    return (out, attrs) if return_attrs else out


def write_ct2_model_bin(path, weights, spec="WhisperSpec", revision=3, aliases=None):
    """Writer for the same layout (used to convert HF checkpoints to a WIS model directory and by the tests)."""
    ids = {np.dtype(np.float32): 0, np.dtype(np.int8): 1, np.dtype(np.int16): 2, np.dtype(np.int32): 3, np.dtype(np.float16): 4}

    def wstr(f, x):
        b = x.encode() + b"\0"
        f.write(struct.pack("<H", len(b)) + b)

    with open(path, "wb") as f:
        f.write(struct.pack("<I", 6))
        wstr(f, spec)
        f.write(struct.pack("<II", revision, len(weights)))
        for name, v in weights.items():
            v = np.asarray(v)
            if v.ndim and not v.flags["C_CONTIGUOUS"]:
                v = np.ascontiguousarray(v)          # (ascontiguousarray would promote a rank-0 attribute to rank 1)
            wstr(f, name)
            f.write(struct.pack("<B", v.ndim) + struct.pack("<" + "I" * v.ndim, *v.shape))
            f.write(struct.pack("<BI", ids[v.dtype], v.nbytes))
            f.write(v.tobytes())
        aliases = aliases or {}
        f.write(struct.pack("<I", len(aliases)))
        for al, tgt in aliases.items():
            wstr(f, al)
            wstr(f, tgt)


# ---- int8_float16: what the engine does to the decoder weights, restated for the oracle -------------------------
DECODER_LINEARS = ("self_attention/linear_0", "self_attention/linear_1", "attention/linear_0", "attention/linear_2", "ffn/linear_0", "ffn/linear_1")


def quantize_rows(w):
    """CTranslate2's per-row int8 scheme as csrc/dec_kernels.hip implements it (float32 arithmetic, round-half-even):
    scale = absmax(row) / 127 (1 for an all-zero row), q = clip(rint(w / scale), -127, 127).  Returns (q int8, scale f32)."""
    w32 = np.asarray(w, np.float32)
    mx = np.abs(w32).max(axis=1)
    scale = np.where(mx > 0, mx / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    inv = (np.float32(1.0) / scale).astype(np.float32)
    q = np.clip(np.rint(w32 * inv[:, None]), -127, 127).astype(np.int8)
    return q, scale


def quantize_folded(w, gamma):
    """What the engine stores for a projection that follows a LayerNorm: the gamma-folded matrix f16(W * gamma) quantised per
    row; returned as the equivalent UNFOLDED float32 matrix (divided by gamma again) so that an oracle applying gamma itself
    reproduces it."""
    g = np.asarray(gamma, np.float32)
    wg = (np.asarray(w, np.float32) * g[None, :]).astype(np.float16)
    q, sc = quantize_rows(wg)
    deq = q.astype(np.float32) * sc[:, None]
    safe = np.where(g != 0, g, np.float32(1.0))
    return np.where(g[None, :] != 0, deq / safe[None, :], np.float32(0.0)).astype(np.float32)


def quantize_decoder_weights(weights):
    """name -> array dict in which every decoder matrix the engine stores as int8 under compute_type="int8_float16" (the six
    linears of each decoder layer and the vocabulary projection) is replaced by its float32 de-quantised value - the matrices
    that follow a LayerNorm (QKV, cross-Q, FFN1, the projection) as the engine quantises them, gamma-folded; the embedding
    LOOKUP table, the cross K/V projection and the whole encoder stay f16, exactly as in the engine."""
    out = dict(weights)
    ln_of = {"self_attention/linear_0": "self_attention/layer_norm", "attention/linear_0": "attention/layer_norm", "ffn/linear_0": "ffn/layer_norm"}
    for name in list(weights):
        if not (name.startswith("decoder/layer_") and name.endswith("/weight")):
            continue
        lin = next((l for l in DECODER_LINEARS if name.endswith(l + "/weight")), None)
        if lin is None:
            continue
        if lin in ln_of:
            gamma = weights[name[:-len(lin + "/weight")] + ln_of[lin] + "/gamma"]
            out[name] = quantize_folded(weights[name], gamma)
        else:
            q, sc = quantize_rows(weights[name])
            out[name] = q.astype(np.float32) * sc[:, None]
    out["decoder/projection/weight"] = quantize_folded(weights["decoder/embeddings/weight"], weights["decoder/layer_norm/gamma"])
    return out


# ---- Hugging Face checkpoints (HF names per transformers modeling_whisper.py) ----------------
def from_hf_state_dict(sd, dtype=np.float16):
    """HF `WhisperForConditionalGeneration` state dict (name -> ndarray) -> CTranslate2 WhisperSpec names.
    q/k/v are fused into linear_0 ([3d, d]; the absent k_proj bias becomes zeros), the cross-attention k/v into
    attention/linear_1 ([2d, d]); Q scaling stays OUT of the weights (CT2 applies it at run time; this engine folds it in
    when it packs the decoder weights, csrc/model.hip)."""
    g = lambda n: np.asarray(sd[n])
    out = {}

    def cast(v):
        return np.ascontiguousarray(np.asarray(v, np.float32).astype(dtype))

    pre = "model." if any(k.startswith("model.") for k in sd) else ""
    d = g(pre + "encoder.conv1.weight").shape[0]
    out["encoder/conv1/weight"], out["encoder/conv1/bias"] = cast(g(pre + "encoder.conv1.weight")), cast(g(pre + "encoder.conv1.bias"))
    out["encoder/conv2/weight"], out["encoder/conv2/bias"] = cast(g(pre + "encoder.conv2.weight")), cast(g(pre + "encoder.conv2.bias"))
    key = pre + "encoder.embed_positions.weight"
    out["encoder/position_encodings/encodings"] = (np.ascontiguousarray(np.asarray(sd[key], np.float32)) if key in sd
                                                   else sinusoids(N_AUDIO_CTX, d))
    zeros = np.zeros(d, np.float32)
    for side in ("encoder", "decoder"):
        L = 1 + max(int(k.split(".")[len(pre.split(".")) + 1]) for k in sd if k.startswith(f"{pre}{side}.layers."))
        for l in range(L):
            p, q = f"{side}/layer_{l}/", f"{pre}{side}.layers.{l}."
            out[p + "self_attention/layer_norm/gamma"], out[p + "self_attention/layer_norm/beta"] = cast(g(q + "self_attn_layer_norm.weight")), cast(g(q + "self_attn_layer_norm.bias"))
            out[p + "self_attention/linear_0/weight"] = cast(np.concatenate([g(q + "self_attn.q_proj.weight"), g(q + "self_attn.k_proj.weight"), g(q + "self_attn.v_proj.weight")]))
            out[p + "self_attention/linear_0/bias"] = cast(np.concatenate([np.asarray(g(q + "self_attn.q_proj.bias"), np.float32), zeros, np.asarray(g(q + "self_attn.v_proj.bias"), np.float32)]))
            out[p + "self_attention/linear_1/weight"], out[p + "self_attention/linear_1/bias"] = cast(g(q + "self_attn.out_proj.weight")), cast(g(q + "self_attn.out_proj.bias"))
            if side == "decoder":
                out[p + "attention/layer_norm/gamma"], out[p + "attention/layer_norm/beta"] = cast(g(q + "encoder_attn_layer_norm.weight")), cast(g(q + "encoder_attn_layer_norm.bias"))
                out[p + "attention/linear_0/weight"], out[p + "attention/linear_0/bias"] = cast(g(q + "encoder_attn.q_proj.weight")), cast(g(q + "encoder_attn.q_proj.bias"))
                out[p + "attention/linear_1/weight"] = cast(np.concatenate([g(q + "encoder_attn.k_proj.weight"), g(q + "encoder_attn.v_proj.weight")]))
                out[p + "attention/linear_1/bias"] = cast(np.concatenate([zeros, np.asarray(g(q + "encoder_attn.v_proj.bias"), np.float32)]))
                out[p + "attention/linear_2/weight"], out[p + "attention/linear_2/bias"] = cast(g(q + "encoder_attn.out_proj.weight")), cast(g(q + "encoder_attn.out_proj.bias"))
            out[p + "ffn/layer_norm/gamma"], out[p + "ffn/layer_norm/beta"] = cast(g(q + "final_layer_norm.weight")), cast(g(q + "final_layer_norm.bias"))
            out[p + "ffn/linear_0/weight"], out[p + "ffn/linear_0/bias"] = cast(g(q + "fc1.weight")), cast(g(q + "fc1.bias"))
            out[p + "ffn/linear_1/weight"], out[p + "ffn/linear_1/bias"] = cast(g(q + "fc2.weight")), cast(g(q + "fc2.bias"))
        out[f"{side}/layer_norm/gamma"], out[f"{side}/layer_norm/beta"] = cast(g(f"{pre}{side}.layer_norm.weight")), cast(g(f"{pre}{side}.layer_norm.bias"))
    out["decoder/embeddings/weight"] = cast(g(pre + "decoder.embed_tokens.weight"))
    out["decoder/position_encodings/encodings"] = cast(g(pre + "decoder.embed_positions.weight"))
    return out


def _read_safetensors(path):
    from safetensors import safe_open
    out = {}
    with safe_open(path, framework="np") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out


def load_hf_dir(path):
    """A Hugging Face Whisper checkpoint directory (`config.json` + `model.safetensors` or its sharded index
    [+ `generation_config.json`]) -> (CT2-named weights, arch dict, decode config)."""
    with open(os.path.join(path, "config.json")) as f:
        hc = json.load(f)
    idx = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    else:
        files = ["model.safetensors"]
    sd = {}
    for fn in files:
        sd.update(_read_safetensors(os.path.join(path, fn)))
    w = from_hf_state_dict(sd)
    gen = {}
    gj = os.path.join(path, "generation_config.json")
    if os.path.exists(gj):
        with open(gj) as f:
            gen = json.load(f)
    cfg = {}
    sup = gen.get("suppress_tokens", hc.get("suppress_tokens"))
    if sup:
        # CT2's converter adds <|translate|>, <|transcribe|> handling through the prompt; the released WIS models carry
        # them in suppress_ids - keep whatever the checkpoint says and add the two task tokens
        cfg["suppress_ids"] = sorted(set(int(t) for t in sup) | {TRANSLATE, TRANSCRIBE})
    beg = gen.get("begin_suppress_tokens", hc.get("begin_suppress_tokens"))
    if beg:
        cfg["suppress_ids_begin"] = [int(t) for t in beg]
    if "lang_to_id" in gen:
        cfg["lang_ids"] = sorted(int(v) for v in gen["lang_to_id"].values())
    return w, arch_from_weights(w, hc.get("decoder_attention_heads")), cfg


def arch_from_weights(w, n_heads=None):
    d = w["decoder/embeddings/weight"].shape[1]
    L = 1 + max(int(k.split("/")[1][len("layer_"):]) for k in w if k.startswith("decoder/layer_") and k.split("/")[1][len("layer_"):].isdigit())
    H = n_heads or d // 64
    if d % 64 or H * 64 != d:
        raise ValueError(f"unsupported Whisper geometry d_model={d}, heads={H}: the engine is built for head_dim 64")
    if w["encoder/conv1/weight"].shape[1] != N_MELS:
        raise ValueError(f"{w['encoder/conv1/weight'].shape[1]} mel bins: only the 80-bin models WIS serves are supported")
    size = next((k for k, v in ARCH.items() if v == (d, L, H)), f"custom-{d}x{L}")
    return dict(size=size, d_model=d, n_layers=L, n_heads=H, n_vocab=int(w["decoder/embeddings/weight"].shape[0]),
                n_audio_ctx=int(w["encoder/position_encodings/encodings"].shape[0]),
                n_text_ctx=int(w["decoder/position_encodings/encodings"].shape[0]), n_mels=N_MELS)


def load_model_dir(path):
    """A Whisper model directory -> (weights, arch, decode config).  Accepted layouts: the CTranslate2 one WIS ships
    (`model.bin` + `config.json` with suppress_ids / suppress_ids_begin / lang_ids; utils.sh:99-108, main.py:341-444) and a
    Hugging Face checkpoint (`model.safetensors`)."""
    if os.path.exists(os.path.join(path, "model.bin")):
        w, attrs = read_ct2_model_bin(os.path.join(path, "model.bin"), return_attrs=True)
        if "encoder/position_encodings/encodings" not in w:
            w["encoder/position_encodings/encodings"] = sinusoids(N_AUDIO_CTX, w["decoder/embeddings/weight"].shape[1])
        w["encoder/position_encodings/encodings"] = np.ascontiguousarray(w["encoder/position_encodings/encodings"], np.float32)
        w.pop("decoder/projection/weight", None)       # alias of the embeddings (tied)
        w = {k: (v if k == "encoder/position_encodings/encodings" else np.ascontiguousarray(v.astype(np.float16)))
             for k, v in w.items() if not k.endswith("weight_scale")}
        cfg = {}
        cj = os.path.join(path, "config.json")
        if os.path.exists(cj):
            with open(cj) as f:
                cj = json.load(f)
            cfg = {k: cj[k] for k in ("suppress_ids", "suppress_ids_begin", "lang_ids") if k in cj}
        heads = attrs.get("decoder/num_heads", attrs.get("encoder/num_heads"))
        return w, arch_from_weights(w, int(np.asarray(heads).reshape(-1)[0]) if heads is not None else None), cfg
    if os.path.exists(os.path.join(path, "model.safetensors")) or os.path.exists(os.path.join(path, "model.safetensors.index.json")):
        return load_hf_dir(path)
    raise FileNotFoundError(f"{path}: neither a CTranslate2 (model.bin) nor a Hugging Face (model.safetensors) Whisper directory")


def convert_hf_to_ct2_dir(hf_dir, out_dir):
    """`ct2-transformers-converter --quantization float16` equivalent for Whisper (utils.sh:104-105): writes model.bin +
    config.json and copies the tokenizer files so `WhisperProcessor.from_pretrained(out_dir)` keeps working (main.py:331-333)."""
    import shutil
    w, a, cfg = load_hf_dir(hf_dir)
    os.makedirs(out_dir, exist_ok=True)
    write_ct2_model_bin(os.path.join(out_dir, "model.bin"), w, aliases={"decoder/projection/weight": "decoder/embeddings/weight"})
    full = dict(suppress_ids=cfg.get("suppress_ids", SUPPRESS_IDS), suppress_ids_begin=cfg.get("suppress_ids_begin", SUPPRESS_IDS_BEGIN),
                lang_ids=cfg.get("lang_ids", LANG_IDS), alignment_heads=[])
    with open(os.path.join(out_dir, "config.json"), "w") as f:
        json.dump(full, f)
    for fn in ("tokenizer.json", "tokenizer_config.json", "preprocessor_config.json", "vocab.json", "merges.txt", "added_tokens.json",
               "special_tokens_map.json", "normalizer.json"):
        if os.path.exists(os.path.join(hf_dir, fn)):
            shutil.copy(os.path.join(hf_dir, fn), os.path.join(out_dir, fn))
    return a
