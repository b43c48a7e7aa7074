"""ctypes binding of libspecdec.so (the C ABI declared in include/specdec.h).

There is no CPU fallback: if the shared object is missing or fails to load the
import raises, and every entry point converts a negative sd_status into the
Python exception the reference raises at the same place (Errors).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SD_LIBSPECDEC") or os.path.join(_HERE, "libspecdec.so")   # (override: kernel-variant experiments)

SD_OK, SD_ERR_INVALID, SD_ERR_NORM_LOGITS, SD_ERR_PROB, SD_ERR_HIP, SD_ERR_CAPACITY = 0, -1, -2, -3, -4, -5
SD_F32, SD_BF16, SD_F16 = 0, 1, 2
SD_NORM_ROUND_BF16, SD_NORM_ROUND_F16, SD_NORM_DT_BF16, SD_NORM_DT_F16 = 1, 2, 16, 32
N_PROFILE_CLASSES = 8
PROFILE_CLASS_NAMES = ["gemm", "attention", "norm_residual", "qkv_rope_append", "activation", "embed",
                       "logits", "other"]


class SdAcceptResult(C.Structure):
    _fields_ = [("n_accepted", C.c_int32), ("n", C.c_int32), ("next_token", C.c_int32), ("flags", C.c_int32),
                ("p_at", C.c_float * 16), ("q_at", C.c_float * 16), ("drafted", C.c_int32 * 16)]


class SdNormRow(C.Structure):
    _fields_ = [("probs_out", C.c_void_p), ("err", C.c_void_p), ("exp_noise", C.c_void_p), ("philox_seed", C.c_uint64),
                ("draw_index", C.c_uint64), ("tok_out", C.c_void_p), ("sample_err", C.c_void_p)]


class SdAcceptItem(C.Structure):
    _fields_ = [("p_hist", C.c_void_p), ("q_hist", C.c_void_p), ("seq", C.c_void_p), ("L", C.c_int32),
                ("r", C.c_void_p), ("exp_noise", C.c_void_p), ("philox_seed", C.c_uint64), ("draw_scan", C.c_uint64),
                ("draw_resample", C.c_uint64), ("res", C.c_void_p), ("err_flags", C.c_void_p), ("n_err", C.c_int32)]


class SdMultiResult(C.Structure):
    _fields_ = [("chosen", SdAcceptResult), ("choice", C.c_int32), ("n_uniform", C.c_int32), ("width", C.c_int32),
                ("gamma", C.c_int32), ("p_at", C.c_float * 256), ("q_at", C.c_float * 256)]


class SdMultiItem(C.Structure):
    _fields_ = [("p_hist", C.c_void_p), ("q_hist", C.c_void_p), ("seq", C.c_void_p)]


class SdBatchStream(C.Structure):
    _fields_ = [("draft", C.c_void_p), ("target", C.c_void_p), ("seq", C.c_void_p), ("q_hist", C.c_void_p),
                ("p_hist", C.c_void_p), ("err_words", C.c_void_p), ("res_dev", C.c_void_p), ("res_host", C.c_void_p),
                ("host_seq", C.c_void_p), ("len", C.c_int32), ("T", C.c_int32), ("ori_eos_cnt", C.c_int32),
                ("draft_len", C.c_int32), ("target_len", C.c_int32), ("seed", C.c_uint64), ("draw", C.c_uint64),
                ("done", C.c_int32), ("calls", C.c_int32), ("acc_len_out", C.c_void_p), ("p_at_out", C.c_void_p),
                ("q_at_out", C.c_void_p)]


class SdBatchItem(C.Structure):
    _fields_ = [("session", C.c_void_p), ("seq", C.c_void_p), ("pos0", C.c_int32), ("n_new", C.c_int32),
                ("n_logits", C.c_int32)]


class SdModelConfig(C.Structure):
    _fields_ = [("arch", C.c_int32), ("dtype", C.c_int32), ("vocab", C.c_int32), ("hidden", C.c_int32),
                ("inter", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
                ("head_dim", C.c_int32), ("max_pos", C.c_int32), ("opt_pre_ln", C.c_int32),
                ("opt_proj_dim", C.c_int32), ("norm_eps", C.c_float), ("logits_bf16_round", C.c_int32),
                ("fused_layout", C.c_int32)]


_VP = C.c_void_p
_VPP = C.POINTER(C.c_void_p)


class SdModelWeights(C.Structure):
    _fields_ = [("embed", _VP), ("pos_embed", _VP), ("project_in", _VP), ("project_out", _VP),
                ("final_norm_w", _VP), ("final_norm_b", _VP), ("lm_head", _VP), ("rope_cos", _VP), ("rope_sin", _VP),
                ("wqkv", _VPP), ("bqkv", _VPP), ("wo", _VPP), ("bo", _VPP), ("w_gate_up", _VPP), ("b_fc1", _VPP),
                ("w_down", _VPP), ("b_fc2", _VPP), ("norm1_w", _VPP), ("norm1_b", _VPP), ("norm2_w", _VPP),
                ("norm2_b", _VPP)]


# every symbol include/specdec.h declares: (name, restype, argtypes)
_F, _I, _L, _U64 = C.c_float, C.c_int, C.c_long, C.c_uint64
SYMBOLS = [
    ("sd_version", _I, []),
    ("sd_last_error", C.c_char_p, []),
    ("sd_norm_probs", _I, [_VP, _I, _I, _L, _F, _I, _F, _I, _VP, _L, _VP, _VP, _VP]),
    ("sd_topk_topp_filter", _I, [_VP, _I, _I, _L, _I, _F, _I, _VP, _L, _VP]),
    ("sd_norm_workspace_bytes", C.c_size_t, [_I]),
    ("sd_norm_sample", _I, [_VP, _I, _F, _I, _F, _I, _VP, _VP, _VP, _U64, _U64, _VP, _VP, _VP, _VP]),
    ("sd_norm_batch", _I, [_VP, _I, _I, _L, _F, _I, _F, _I, C.POINTER(SdNormRow), _I, _VP, _VP]),
    ("sd_accept_batch", _I, [C.POINTER(SdAcceptItem), _I, _L, _I, _I, _I, _VP]),
    ("sd_accept_multi", _I, [C.POINTER(SdMultiItem), _I, _L, _I, _I, _VP, _U64, _U64, _VP, _VP]),
    ("sd_multi_resample", _I, [_VP, _VP, _L, _I, _VP, _I, _VP, _U64, _U64, _VP, _I, _VP]),
    ("sd_sample", _I, [_VP, _I, _VP, _U64, _U64, _VP, _VP, _I, _VP]),
    ("sd_philox_exp", _I, [_U64, _U64, _I, _VP, _VP]),
    ("sd_philox_uniform", _I, [_U64, _U64, _I, _VP, _VP]),
    ("sd_max_fn", _I, [_VP, _VP, _I, _VP, _I, _VP]),
    ("sd_accept_scan", _I, [_VP, _VP, _L, _VP, _I, _I, _VP, _U64, _U64, _VP, _VP]),
    ("sd_resample", _I, [_VP, _VP, _L, _I, _VP, _I, _I, _VP, _U64, _U64, _VP, _VP, _I, _VP]),
    ("sd_model_create", _I, [C.POINTER(SdModelConfig), C.POINTER(SdModelWeights), C.POINTER(_VP)]),
    ("sd_model_destroy", _I, [_VP]),
    ("sd_model_max_rows", _I, [_VP]),
    ("sd_pack_weight_bf16", _I, [_VP, _VP, _I, _I, _VP]),
    ("sd_pack_activation_bf16", _I, [_VP, _VP, _I, _I, _VP]),
    ("sd_gemm_bf16", _I, [_VP, _VP, _I, _I, _I, _I, _VP, C.c_size_t, _VP, C.POINTER(C.c_int), _VP]),
    ("sd_session_kv_bytes", C.c_size_t, [_VP, _I]),
    ("sd_session_scratch_bytes", C.c_size_t, [_VP, _I]),
    ("sd_session_create", _I, [_VP, _I, _I, _VP, _VP, C.POINTER(_VP)]),
    ("sd_session_destroy", _I, [_VP]),
    ("sd_session_set_kv_fp8", _I, [_VP, _VP]),
    ("sd_session_forward", _I, [_VP, _VP, _I, _I, _I, _VP, _L, _VP]),
    ("sd_session_forward_tree", _I, [_VP, _VP, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), _I, _I, _VP, _L, _VP]),
    ("sd_session_compact_kv", _I, [_VP, _I, _VP, _I, _VP]),
    ("sd_session_fused_status", _I, [_VP, _VP]),
    ("sd_session_test_skew_wait", _I, [_VP, _I]),
    ("sd_session_ao_stamps", _I, [_VP, _VP, _I]),
    ("sd_cand_list_bytes", C.c_size_t, [_I]),
    ("sd_norm_probs_lists", _I, [_VP, _I, _I, _L, _F, _I, _F, _I, _VP, _L, _VP, _VP, _VP, _VP]),
    ("sd_accept_resample", _I, [_VP, _VP, _L, _I, _VP, _I, _I, _VP, _U64, _U64, _U64, _VP, _VP, _I, _I, _VP, _VP]),
    ("sd_spec_batch_generate", _I, [_VP, _I, _I, C.c_float, _I, C.c_float, _I, C.c_long, _I, C.c_uint64, _VP, _I, _I, _VP,
                                    C.c_long, _VP, C.c_long, _VP, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP]),
    ("sd_spec_generate", _I, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, C.c_uint64, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP,
                              _VP, _VP, _VP]),
    ("sd_batch_forward", _I, [C.POINTER(SdBatchItem), _I, _VP, _L, _VP]),
    ("sd_batch_prefill", _I, [C.POINTER(SdBatchItem), _I, _VP]),
    ("sd_spec_create", _I, [_VP, _VP, _I, _F, _I, _F, _VP, _VP, _VP, _L, _VP, _L, _VP, _L, _VP, _VP, _VP, C.POINTER(_VP)]),
    ("sd_spec_destroy", _I, [_VP]),
    ("sd_spec_iteration", _I, [_VP, _I, _I, _I, _U64, _U64, _U64, _U64, _U64, _VP, _VP, _VP, _VP]),
    ("sd_spec_timing", _I, [_VP, _I]),
    ("sd_spec_last_times", _I, [_VP, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("sd_tp_unique_id", _I, [_VP]),
    ("sd_tp_create_rccl", _I, [_I, _I, _VP, C.POINTER(_VP)]),
    ("sd_tp_create_loopback", _I, [_I, C.POINTER(_VP)]),
    ("sd_tp_destroy", _I, [_VP]),
    ("sd_session_set_tp", _I, [_VP, _VP]),
    ("sd_comm_probe", _I, []),
    ("sd_comm_unique_id", _I, [_VP]),
    ("sd_comm_init", _I, [_I, _I, _VP, C.POINTER(_VP)]),
    ("sd_comm_all_gather_tokens", _I, [_VP, _VP, _VP, _I, _I, _VP]),
    ("sd_comm_rank", _I, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("sd_comm_destroy", _I, [_VP]),
    ("sd_profile_enable", _I, [_VP, _I]),
    ("sd_profile_read", _I, [_VP, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m llmspeculativesampling_amd._build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the HIP path.")
    # torch first: it brings its own copy of the HIP runtime (torch/lib/libamdhip64.so), and libspecdec.so must bind to THAT
    # instance - loaded before torch it would pull in /opt/rocm's copy, the process would hold two runtimes, and the first
    # kernel launch on torch's device pointers would fail with "no ROCm-capable device is detected"
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class SpecDecError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    """Negative sd_status -> the reference's exception at that point."""
    if rc == SD_OK:
        return
    msg = lib.sd_last_error().decode(errors="replace")
    if rc == SD_ERR_NORM_LOGITS:
        raise RuntimeError("norm logits error")          # reference utils.py:207
    if rc == SD_ERR_PROB:
        raise RuntimeError("prob error")                 # reference utils.py:224
    if rc == SD_ERR_CAPACITY:
        raise SpecDecError(f"{what}: capacity exceeded: {msg}")
    if rc == SD_ERR_INVALID:
        raise ValueError(f"{what}: {msg}")
    raise SpecDecError(f"{what}: HIP failure: {msg}")
