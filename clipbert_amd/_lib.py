"""ctypes binding of libclipbert_hip.so (the C ABI declared in include/clipbert_hip.h).

There is NO fallback: if the shared library is missing or fails to load, importing any op raises.
Build it with ``python -m clipbert_amd.build`` (or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libclipbert_hip.so")

CB_F32, CB_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH = 0, 1, 2, 3
SPLITK_WS_COUNTER_BYTES = 65536                     # CB_SPLITK_WS_COUNTER_BYTES: zeroed tail of cb_gemm_desc.splitk_ws (arrival counters)
ACT_GELU_SAVE_GRAD, ACT_SAVED_GRAD = 4, 5          # cb_gemm only: GELU with C2 = gelu'(pre); backward multiplies by that stored derivative
ROWK, ROWK_GATHER, KROW, KROW_TAPS, KROW_GATHER = 0, 1, 2, 3, 4
(HP_LR, HP_BETA1, HP_BETA2, HP_EPS, HP_WD, HP_BC1, HP_BC2, HP_MAX_NORM, HP_GRAD_SCALE, HP_SKIP, HP_COUNT) = range(11)

vp, i32, i64, f32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("M", i32), ("N", i32), ("K", i32), ("a_mode", i32), ("b_mode", i32),
        ("A", vp), ("B", vp), ("lda", i64), ("ldb", i64), ("a_tab", vp), ("b_tab", vp),
        ("R", i32), ("S", i32), ("Cin", i32), ("H", i32), ("W", i32),
        ("sH", i64), ("sW", i64), ("flip_taps", i32), ("schedule", i32),
        ("C", vp), ("ldc", i64), ("c_rowmap", vp), ("c_f32", i32), ("accumulate", i32),
        ("split_k", i32), ("act", i32), ("scale", vp), ("shift", vp), ("residual", vp), ("ldr", i64),
        ("relu_after", i32), ("zero_fill_pitch", i32), ("mask", vp), ("ldm", i64), ("C2", vp), ("ldc2", i64),
        ("alpha", f32), ("dropout_p", f32), ("dropout_seed", u64), ("dropout_seed_ptr", vp), ("tile", i32),
        ("xcd_order", i32), ("a_bytes", i64), ("b_bytes", i64), ("gelu_grad_pre", vp), ("ld_gelu", i64), ("a_rowsum", vp),
        ("batch", i32), ("relu_bwd", i32), ("batch_stride_a", i64), ("batch_stride_b", i64), ("batch_stride_c", i64),
        ("batch_stride_rowsum", i64), ("post_scale", vp), ("post_scale2", vp), ("splitk_ws", vp), ("splitk_ws_bytes", i64),
        ("sq_slots", vp), ("sq_slots_n", i64),
    ]


class Res2Desc(C.Structure):
    _fields_ = [("x", vp), ("out", vp), ("w1", vp), ("w2", vp), ("w3", vp), ("wsc", vp),
                ("scale1", vp), ("shift1", vp), ("scale2", vp), ("shift2", vp), ("scale3", vp), ("shift3", vp),
                ("scale_sc", vp), ("shift_sc", vp), ("N", i32), ("H", i32), ("W", i32), ("cin", i32)]


_SIGNATURES = {
    "cb_gemm": [C.POINTER(GemmDesc), vp],
    "cb_res2_block": [C.POINTER(Res2Desc), vp],
    "cb_stem_pool": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "cb_stem_pool_u8": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "cb_gemm_plan": [vp, i32, vp],
    "cb_gemm_group": [vp, i32, vp],
    "cb_gemm_workspace_bytes": [vp, vp],
    "cb_build_pixel_table": [vp, i32, i32, i32, i32, i32, i64, i64, i64, vp],
    "cb_stem_pack": [i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "cb_image_norm": [vp, vp, vp, vp, i64, i64, vp],
    "cb_maxpool_fwd": [i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "cb_maxpool2_bwd": [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "cb_relu_scale_bwd": [i32, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp],
    "cb_layernorm_fwd": [i32, vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, i32, i32, vp],
    "cb_layernorm_bwd": [i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, f32, u64, vp, i32, i32, i32, vp],
    "cb_layernorm_bwd_part": [i32, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, vp, f32, u64, vp, i32, i32, i32, vp],
    "cb_ln_partials_reduce": [vp, vp, vp, vp, i32, i32, i32, vp],
    "cb_comm_unique_id": [vp],
    "cb_comm_init": [i32, i32, vp],
    "cb_comm_info": [vp, vp],
    "cb_allreduce_bucket": [vp, i64, i32, vp],
    "cb_reduce_scatter_bucket": [vp, vp, i64, i32, vp],
    "cb_allgather_bucket": [vp, vp, i64, i32, vp],
    "cb_broadcast_bucket": [vp, i64, i32, i32, vp],
    "cb_comm_destroy": [],
    "cb_text_embed_fwd": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, i32, vp],
    "cb_visual_embed_fwd": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32,
                            i32, i32, i32, f32, vp, vp],
    "cb_text_embed_bwd": [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i32, vp],
    "cb_head_loss": [i32, vp, vp, vp, vp, vp, i64, i32, f32, vp],
    "cb_retrieval_scores": [vp, vp, i64, i32, vp],
    "cb_sq_sum_fold": [vp, vp, i32, vp, i64, vp, vp, i32, vp],
    "cb_mean_fwd": [vp, i64, vp, vp],
    "cb_mean_bwd": [vp, i64, vp, vp],
    "cb_counter_add": [vp, i64, vp],
    "cb_zero": [vp, i64, vp],
    "cb_zero_ranges": [vp, vp, i32, vp],
    "cb_visual_embed_bwd": [i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "cb_attention_fwd": [i32, vp, vp, vp, vp, i32, i32, i32, f32, u64, vp, vp],
    "cb_attention_bwd": [i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, u64, vp, vp],
    "cb_cross_entropy": [vp, i64, vp, vp, vp, vp, i64, i32, i64, vp],
    "cb_colsum": [i32, vp, i64, vp, i64, i32, vp],
    "cb_cast": [i32, vp, i32, vp, i64, vp],
    "cb_act_bwd": [i32, i32, vp, vp, vp, i64, vp],
    "cb_adamw": [vp, vp, vp, vp, vp, i64, vp, vp, vp],
    "cb_dropout": [i32, vp, vp, i64, f32, u64, vp, vp],
    "cb_clip_aggregate_fwd": [vp, i32, i64, i32, vp, vp, vp],
    "cb_clip_aggregate_bwd": [vp, vp, vp, vp, i32, i64, i32, vp, vp],
    "cb_lse_loss": [vp, vp, i32, i32, i32, vp, vp, vp, vp],
    "cb_sq_sum": [vp, i64, vp, vp],
    "cb_sq_sum_det": [vp, i64, vp, vp, i32, vp],
    "cb_sq_sum_det_bf16": [vp, i64, vp, vp, i32, vp],
    "cb_adamw_g16": [vp, vp, vp, vp, vp, i64, vp, vp, vp],
    "cb_elu_bn1d_fwd": [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp],
    "cb_elu_bn1d_bwd": [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES) + ("cb_last_error", "cb_version")


def bind(cdll, strict: bool = True):
    """Attach argtypes/restype for every entry point of the C ABI; raises if a symbol is missing."""
    for name, args in _SIGNATURES.items():
        if not strict and not hasattr(cdll, name):
            continue
        fn = getattr(cdll, name)
        fn.argtypes = args
        fn.restype = C.c_int
    cdll.cb_last_error.argtypes = []
    cdll.cb_last_error.restype = C.c_char_p
    cdll.cb_version.argtypes = []
    cdll.cb_version.restype = C.c_int
    return cdll


_LIB = None


def load(path: str = LIB_PATH):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP library is required (no fallback path exists). "
            "Build it with `python -m clipbert_amd.build`.")
    import torch  # noqa: F401  (its bundled HIP runtime must be loaded FIRST: the library then binds the same libamdhip64)
    return bind(C.CDLL(path))


def get():
    global _LIB
    if _LIB is None:
        variant = os.environ.get("CB_LIB_VARIANT")        # diagnostics: a copy built by `python -m clipbert_amd.build --variant NAME ...`
        _LIB = load(os.path.join(os.path.dirname(LIB_PATH), f"libclipbert_hip_{variant}.so")) if variant else load()
    return _LIB


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = get().cb_last_error().decode(errors="replace")
        raise RuntimeError(f"{what or 'libclipbert_hip'} failed: {msg}")
