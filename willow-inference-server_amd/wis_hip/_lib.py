"""ctypes binding of libwis_hip.so (include/wis_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or no gfx950 device is
visible, every compute entry point raises.  Build the library with
`python willow-inference-server_amd/build.py` (or `__graft_entry__.build()`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WIS_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libwis_hip.so")     # WIS_LIB_PATH: a tuning build (build.py --variant)

WIS_OK = 0
WIS_IN_MEL_HOST, WIS_IN_MEL_DEV, WIS_IN_PCM_HOST, WIS_IN_PCM_DEV = 0, 1, 2, 3
WIS_DT_F32, WIS_DT_F16 = 0, 1
N_SAMPLES, N_FRAMES, N_MELS = 480000, 3000, 80


class WisError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libwis_hip error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d_model", "n_heads", "n_enc_layers", "n_dec_layers", "n_vocab", "n_audio_ctx",
                                          "n_text_ctx", "n_mels", "max_batch", "max_beam", "eot", "sot", "no_timestamps",
                                          "no_speech")] + [
        ("suppress_ids", C.POINTER(C.c_int32)), ("n_suppress", C.c_int32),
        ("suppress_ids_begin", C.POINTER(C.c_int32)), ("n_suppress_begin", C.c_int32),
        ("lang_ids", C.POINTER(C.c_int32)), ("n_lang", C.c_int32), ("decoder_weight_bits", C.c_int32)]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("rank", C.c_int32), ("shape", C.c_int64 * 4), ("offset", C.c_uint64)]


class GenOpts(C.Structure):
    _fields_ = [("input_kind", C.c_int32), ("beam_size", C.c_int32), ("max_new_tokens", C.c_int32), ("length_penalty", C.c_float),
                ("patience", C.c_float), ("suppress_blank", C.c_int32), ("suppress_default", C.c_int32),
                ("fixed_new_tokens", C.c_int32), ("queue_depth", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("logmel_ms", C.c_float), ("encoder_ms", C.c_float), ("crosskv_ms", C.c_float), ("prefill_ms", C.c_float),
                ("decode_ms", C.c_float), ("total_ms", C.c_float), ("decode_steps", C.c_int32), ("decode_steps_needed", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/wis_hip.h declares: (name, restype, argtypes)
_vp, _i, _sz, _i64 = C.c_void_p, C.c_int, C.c_size_t, C.c_int64
_fp = C.POINTER(C.c_float)
SYMBOLS = [
    ("wis_version", _i, []),
    ("wis_last_error", C.c_char_p, []),
    ("wis_device_count", _i, []),
    ("wis_supported_compute_types", _i, [_i, C.c_char_p, _sz]),
    ("wis_audio_decode", _i, [_vp, _sz, C.POINTER(_fp), C.POINTER(_i64), C.POINTER(_i), C.POINTER(_i)]),
    ("wis_audio_free", None, [_fp]),
    ("wis_logmel", _i, [_i, _vp, _i64, C.POINTER(_i64), _i, _i, _vp, _i]),
    ("wis_melstream_create", _i, [_i, C.POINTER(_vp)]),
    ("wis_melstream_reset", _i, [_vp]),
    ("wis_melstream_feed", _i, [_vp, _vp, _i64]),
    ("wis_melstream_finish", _i, [_vp, _vp, C.POINTER(_vp)]),
    ("wis_melstream_samples", _i64, [_vp]),
    ("wis_melstream_tiles_done", _i, [_vp]),
    ("wis_melstream_destroy", None, [_vp]),
    ("wis_model_create", _i, [C.POINTER(Config), _vp, _sz, _i, C.POINTER(Tensor), _i, _i, C.POINTER(_vp)]),
    ("wis_model_destroy", None, [_vp]),
    ("wis_model_device_bytes", _sz, [_vp]),
    ("wis_model_clone", _i, [_vp, C.POINTER(_vp)]),
    ("wis_generate", _i, [_vp, _vp, _i, C.POINTER(C.c_int32), _i, C.POINTER(GenOpts), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _fp]),
    ("wis_generate_draft", _i, [_vp, _vp, C.POINTER(C.c_int32), _i, C.POINTER(GenOpts), C.POINTER(C.c_int32), _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _fp,
                                C.POINTER(C.c_int32)]),
    ("wis_generate_draft_beam", _i, [_vp, _vp, C.POINTER(C.c_int32), _i, C.POINTER(GenOpts), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), _fp, C.POINTER(C.c_int32)]),
    ("wis_last_trajectory", _i, [_vp, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i, C.POINTER(C.c_int32)]),
    ("wis_detect_language", _i, [_vp, _vp, _i, _i, _fp]),
    ("wis_debug_encode", _i, [_vp, _vp, _i, _i, _fp]),
    ("wis_debug_logits", _i, [_vp, _vp, _i, _i, C.POINTER(C.c_int32), _i, _fp]),
    ("wis_debug_logits_rows", _i, [_vp, _vp, _i, _i, C.POINTER(C.c_int32), _i, _i, _fp]),
    ("wis_debug_tree_logits", _i, [_vp, _vp, _i, C.POINTER(C.c_int32), _i, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i, _fp]),
    ("wis_debug_search", _i, [_vp, _vp, _i, _i, C.POINTER(GenOpts), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _fp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("wis_debug_handoff", _i, [_vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    ("wis_last_timing", _i, [_vp, C.POINTER(Timing)]),
    ("wis_debug_phase_cycles", _i, [_vp, _i, _i, _i, C.POINTER(C.c_uint64)]),
    ("wis_debug_timeline", _i, [_vp, _i, _i, _i, _i, C.POINTER(C.c_uint64), _i]),
    ("wis_debug_sampling_cycles", _i, [_vp, C.POINTER(C.c_uint64)]),
    ("wis_bench_weight_stream", _i, [_vp, _i, _i, _fp, C.POINTER(_i), C.POINTER(C.c_double)]),
    ("wis_dev_alloc", _i, [_i, _sz, C.POINTER(_vp)]),
    ("wis_dev_free", _i, [_i, _vp]),
    ("wis_dev_h2d", _i, [_i, _vp, _vp, _sz]),
    ("wis_dev_d2h", _i, [_i, _vp, _vp, _sz]),
    ("wis_dev_sync", _i, [_i]),
    ("wis_dev_copy_peer", _i, [_i, _vp, _i, _vp, _sz]),
    ("wis_op_gemm", _i, [_i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    ("wis_op_layernorm", _i, [_i, _vp, _vp, _vp, _vp, _i, _i]),
    ("wis_op_enc_attention", _i, [_i, _vp, _vp, _vp, _i, _i, _i, _i]),
    ("wis_op_gemv", _i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    ("wis_op_dec_self_attn", _i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    ("wis_op_dec_cross_attn", _i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    ("wis_op_dec_cross_attn_folded", _i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
]

_lib = None


def load():
    """Load libwis_hip.so (once) and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WisError(-5, f"{LIB_PATH} is not built (run `python willow-inference-server_amd/build.py`); "
                           "there is no CPU fallback for the ASR path")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.wis_version() != 1:
        raise WisError(-6, "ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != WIS_OK:
        raise WisError(rc, (load().wis_last_error() or b"").decode(errors="replace"))


def device_count():
    return load().wis_device_count()


def require_gpu():
    n = device_count()
    if n < 1:
        raise WisError(-5, "no HIP device visible: the wis_hip ASR path needs a gfx950 GPU (no CPU fallback)")
    return n


def ptr(a):
    """void* of a C-contiguous numpy array."""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class DevBuf:
    """A raw device allocation (tests / bench / weight broadcast plumbing)."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(load().wis_dev_alloc(device, self.nbytes, C.byref(p)))
        self.ptr = p

    @classmethod
    def from_numpy(cls, a, device=0):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes, device)
        check(load().wis_dev_h2d(device, b.ptr, ptr(a), a.nbytes))
        return b

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(load().wis_dev_d2h(self.device, ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            load().wis_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
