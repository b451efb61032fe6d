"""ctypes binding of the C ABI in ``include/p3v.h`` (``libp3v.so``).

The library is the product: there is NO fallback.  If it is missing or fails to
load, every op raises -- a silent PyTorch path would void the parity claims.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P3V_LIB") or os.path.join(_HERE, "libp3v.so")     # P3V_LIB: debug builds only

P3V_OK = 0
(EPI_NONE, EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_GELU, EPI_BIAS_RESID_F32, EPI_RESID_BF16, EPI_SILU_MUL, EPI_PATCH,
 EPI_F32) = range(9)
DECODE_MAX_L = 16

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class Props(C.Structure):
    _fields_ = [("cu_count", i32), ("lds_per_cu", i32), ("wave_size", i32), ("clock_khz", i32),
                ("mem_clock_khz", i32), ("mem_bus_bits", i32), ("hbm_bytes", i64), ("arch", C.c_char * 32)]


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("W", vp), ("out", vp), ("bias", vp), ("resid", vp), ("pos", vp),
                ("M", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldw", i32), ("ldo", i32),
                ("epilogue", i32), ("patches_per_img", i32), ("ws", vp), ("ws_bytes", i64)]


class QkvSplit(C.Structure):
    _fields_ = [("cos_t", vp), ("sin_t", vp), ("q_out", vp), ("k_dst", vp), ("v_dst", vp),
                ("B", i32), ("L", i32), ("n_heads", i32), ("n_kv", i32), ("hd", i32),
                ("past", i32), ("dst_t", i32), ("dst_off_is_past", i32), ("tab_t", i32), ("tab_div", i32), ("q_scale", f32)]


class GemvArgs(C.Structure):
    _fields_ = [("x", vp), ("W", vp), ("out", vp), ("resid", vp), ("norm_w", vp), ("norm_eps", f32),
                ("M", i32), ("N", i32), ("K", i32), ("epilogue", i32)]


class GemvStep(C.Structure):
    _fields_ = [("tok", vp), ("embed_table", vp), ("vocab", i32), ("x_out", vp),
                ("cos_t", vp), ("sin_t", vp), ("cos_out", vp), ("sin_out", vp), ("tab_t", i32), ("half_dim", i32),
                ("next_tok", vp), ("tok_out", vp), ("history", vp), ("d_step", vp), ("ticket", vp), ("amax_ws", vp), ("max_steps", i32),
                ("d_past", vp)]


GEMV_STEP_WS_BYTES = 1024 * 8 + 9 * 128     # P3V_GEMV_STEP_WS_BYTES (candidate records + nine arrival counters, zero-initialised)


class GemvF8Args(C.Structure):
    _fields_ = [("x", vp), ("W", vp), ("w_scale", vp), ("out", vp), ("resid", vp), ("norm_w", vp), ("norm_eps", f32),
                ("M", i32), ("N", i32), ("K", i32), ("epilogue", i32)]


class GemmF8Args(C.Structure):
    _fields_ = [("A", vp), ("a_scale", vp), ("W", vp), ("w_scale", vp), ("out", vp), ("resid", vp),
                ("M", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldw", i32), ("ldo", i32), ("epilogue", i32)]


class AttnArgs(C.Structure):
    _fields_ = [("q", vp), ("k_past", vp), ("v_past", vp), ("k_new", vp), ("v_new", vp), ("out", vp),
                ("pad_len", vp), ("d_past", vp), ("ws", vp),
                ("B", i32), ("L", i32), ("n_heads", i32), ("n_kv", i32), ("hd", i32),
                ("past", i32), ("past_t", i32), ("past_div", i32), ("new_t", i32), ("pad_div", i32),
                ("causal", i32), ("scale", f32), ("n_split", i32), ("new_is_cache", i32), ("q_prescaled", i32)]


class GemvQ4Args(C.Structure):
    _fields_ = [("x", vp), ("W", vp), ("sb", vp), ("out", vp), ("resid", vp), ("norm_w", vp), ("norm_eps", f32),
                ("M", i32), ("N", i32), ("K", i32), ("epilogue", i32)]


class AttnDecArgs(C.Structure):
    _fields_ = [("qkv", vp), ("cos_t", vp), ("sin_t", vp), ("k_cache", vp), ("v_cache", vp), ("out", vp),
                ("pad_len", vp), ("d_past", vp), ("ws", vp),
                ("B", i32), ("L", i32), ("n_heads", i32), ("n_kv", i32), ("hd", i32), ("past", i32),
                ("cache_t", i32), ("rope_bstride", i32), ("n_split", i32), ("scale", f32), ("merge_in_launch", i32),
                ("o_proj_w", vp), ("o_proj_x", vp), ("o_rearm", vp), ("o_n", i32),       # optional fused o_proj + residual
                ("o_proj_sb", vp)]                                                       #   ... on 4-bit group-64 weights


class AttnDecQ8Args(C.Structure):
    _fields_ = [("qkv", vp), ("cos_t", vp), ("sin_t", vp), ("k8", vp), ("v8t", vp), ("k_scale", vp), ("v_scale", vp),
                ("out", vp), ("pad_len", vp), ("d_past", vp), ("ws", vp),
                ("B", i32), ("L", i32), ("n_heads", i32), ("n_kv", i32), ("hd", i32), ("past", i32),
                ("cache_t", i32), ("rope_bstride", i32), ("n_split", i32), ("scale", f32), ("merge_in_launch", i32),
                ("o_proj_w8", vp), ("o_proj_scale", vp), ("o_proj_x", vp), ("o_rearm", vp), ("o_n", i32)]   # optional fused o_proj (e4m3)


# name -> (restype, argtypes); must list every symbol include/p3v.h declares
SIGNATURES = {
    "p3v_version": (i32, []),
    "p3v_device_props": (i32, [i32, C.POINTER(Props)]),
    "p3v_strerror": (C.c_char_p, [i32]),
    "p3v_set_tuning": (i32, [C.c_char_p, i32]),
    "p3v_get_tuning": (i32, [C.c_char_p, C.POINTER(i32)]),
    "p3v_embed_gather": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "p3v_rmsnorm": (i32, [vp, vp, vp, i32, i32, f32, vp]),
    "p3v_layernorm": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "p3v_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "p3v_gemm_ws_bytes": (i64, [i32, i32, i32, i32]),
    "p3v_gemm_rows_slices": (i32, [i32, i32, i32, i32]),
    "p3v_gemv": (i32, [C.POINTER(GemvArgs), vp]),
    "p3v_gemv_step": (i32, [C.POINTER(GemvArgs), C.POINTER(GemvStep), vp]),
    "p3v_gemv_fp8_step": (i32, [C.POINTER(GemvF8Args), C.POINTER(GemvStep), vp]),
    "p3v_gemv_q4_step": (i32, [C.POINTER(GemvQ4Args), C.POINTER(GemvStep), vp]),
    "p3v_gemv_fp8": (i32, [C.POINTER(GemvF8Args), vp]),
    "p3v_dequant_fp8": (i32, [vp, vp, vp, i32, i32, vp]),
    "p3v_gemm_fp8": (i32, [C.POINTER(GemmF8Args), vp]),
    "p3v_quant_fp8_rows": (i32, [vp, vp, f32, vp, vp, i32, i32, vp]),
    "p3v_rope_table": (i32, [vp, vp, f32, vp, vp, i32, i32, vp]),
    "p3v_rope_kv_append": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, f32, vp]),
    "p3v_attention": (i32, [C.POINTER(AttnArgs), vp]),
    "p3v_attention_decode": (i32, [C.POINTER(AttnDecArgs), vp]),
    "p3v_attention_decode_can_fuse_oproj": (i32, [i32, i32, i32, i32, i32, i32, i32, i32]),
    "p3v_attention_decode_fused_role": (i32, [i32, i32, i32, i32, i32, C.POINTER(i32)]),
    "p3v_kv_quantize": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "p3v_kv_dequantize": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "p3v_gemm_qkv": (i32, [vp, vp, vp]),
    "p3v_gemm_resid_norm": (i32, [C.POINTER(GemmArgs), vp, f32, vp, vp]),
    "p3v_kv_quantize_mlx4": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "p3v_attention_decode_q8": (i32, [C.POINTER(AttnDecQ8Args), vp]),
    "p3v_attention_decode_q8_can_fuse_oproj": (i32, [i32, i32, i32, i32, i32, i32, i32, i32]),
    "p3v_stage_rope": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "p3v_attention_ws_bytes": (i64, [i32, i32, i32, i32, i32]),
    "p3v_im2col_patches": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "p3v_clip_cls_rows": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "p3v_hd_merge": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "p3v_argmax": (i32, [vp, vp, i32, i32, i64, vp]),
    "p3v_log_softmax": (i32, [vp, vp, i32, i32, vp]),
    "p3v_topk": (i32, [vp, vp, i32, i32, i32, i64, vp]),
    "p3v_add_i32": (i32, [vp, i32, i32, vp]),
    "p3v_store_token": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "p3v_gemv_q4": (i32, [vp, vp]),
    "p3v_dequant_q4": (i32, [vp, vp, vp, i32, i32, vp]),
    "p3v_resample_u8": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]),
    "p3v_hd_preprocess": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "p3v_lora_down": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "p3v_lora_up": (i32, [vp, vp, vp, f32, i32, vp, vp, i32, i32, i32, vp]),
    "p3v_step_begin": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "p3v_step_end": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "p3v_graph_begin": (i32, [vp]),
    "p3v_graph_end": (i32, [vp, C.POINTER(vp)]),
    "p3v_graph_launch": (i32, [vp, vp]),
    "p3v_graph_destroy": (i32, [vp]),
    "p3v_event_create": (i32, [C.POINTER(vp)]),
    "p3v_event_record": (i32, [vp, vp]),
    "p3v_event_elapsed_ms": (i32, [vp, vp, C.POINTER(f32)]),
    "p3v_event_destroy": (i32, [vp]),
}

_lib = None


def lib():
    """Load libp3v.so once; raise loudly when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or csrc/build.sh). "
                "There is no CPU/PyTorch fallback for the hot path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the ABI and the header drift apart
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


ERR_UNSUPPORTED = -95        # P3V_ERR_UNSUPPORTED (include/p3v.h)


def check(code, what=""):
    if code != P3V_OK:
        msg = lib().p3v_strerror(code).decode()
        raise RuntimeError(f"p3v call {what} failed: {code} ({msg})")
