"""ctypes binding of libctrlhip.so (the C ABI declared in include/ctrl_hip.h).

There is no fallback: if the shared library is missing or fails to load, importing the product path
raises.  PyTorch is used only for device memory and streams; every tensor crosses this boundary as a
raw device pointer.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libctrlhip.so")

F32, F16, BF16 = 0, 1, 2


class IGemmSeg(C.Structure):
    _fields_ = [("out", C.c_void_p), ("ld", C.c_int64), ("col_begin", C.c_int32), ("ncols", C.c_int32),
                ("fmt", C.c_int32), ("dtype", C.c_int32), ("L", C.c_int32), ("pad_", C.c_int32), ("img_map", C.c_void_p)]


class IGemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("mode", C.c_int32), ("Cin", C.c_int32), ("taps", C.c_int32),
                ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
                ("stride", C.c_int32), ("up", C.c_int32), ("F", C.c_int32), ("HW", C.c_int32), ("t_pad", C.c_int32),
                ("W", C.c_void_p), ("M", C.c_int32), ("Nout", C.c_int32), ("Ktot", C.c_int32), ("pad1_", C.c_int32),
                ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("rowvec_ld", C.c_int32), ("rows_per_img", C.c_int32),
                ("res", C.c_void_p), ("ldres", C.c_int64), ("scale", C.c_float), ("geglu", C.c_int32),
                ("nseg", C.c_int32), ("act", C.c_int32), ("res_f32", C.c_int32), ("a_split", C.c_int32),
                ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64), ("splitk_tickets", C.c_void_p),
                ("out16", C.c_void_p), ("ld16", C.c_int64),
                ("blend_mix", C.c_void_p), ("blend_x", C.c_void_p), ("ld_blend", C.c_int64), ("blend_f32", C.c_int32), ("out16_lo_off", C.c_int32), ("scale2", C.c_float), ("scale2_from", C.c_int32),
                ("res_up", C.c_int32), ("scale2_to", C.c_int32),
                ("seg", IGemmSeg * 3), ("nonfinite", C.c_void_p)]


class FfnDesc(C.Structure):
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int64), ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2p", C.c_void_p), ("out", IGemmDesc)]


class AttnDesc(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("ldq", C.c_int64), ("K", C.c_void_p), ("ldk", C.c_int64),
                ("Vt", C.c_void_p), ("Lkpad", C.c_int32), ("kvB", C.c_int32),
                ("O", C.c_void_p), ("ldo", C.c_int64),
                ("B", C.c_int32), ("heads", C.c_int32), ("D", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32),
                ("scale", C.c_float), ("k_prescaled", C.c_int32), ("pad0_", C.c_int32)]


class TAttnDesc(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("ld", C.c_int64), ("O", C.c_void_p), ("ldo", C.c_int64),
                ("Bc", C.c_int32), ("F", C.c_int32), ("HW", C.c_int32), ("heads", C.c_int32),
                ("scale", C.c_float), ("Fq", C.c_int32), ("KV", C.c_void_p), ("ldkv", C.c_int64),
                ("Fl", C.c_int32), ("pad0_", C.c_int32)]


class TensorRef(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 6)]


class ControlNetConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("conditioning_channels", C.c_int32),
                ("block_out_channels", C.c_int32 * 4), ("down_block_has_attn", C.c_int32 * 4),
                ("layers_per_block", C.c_int32), ("num_attention_heads", C.c_int32),
                ("cross_attention_dim", C.c_int32), ("cond_embed_channels", C.c_int32 * 4),
                ("norm_eps", C.c_float)]


class AdapterConfig(C.Structure):
    _fields_ = [("backbone_sdxl", C.c_int32), ("num_blocks", C.c_int32), ("num_adapters_per_location", C.c_int32),
                ("cross_attention_dim", C.c_int32),
                ("add_spatial_resnet", C.c_int32), ("add_temporal_resnet", C.c_int32),
                ("add_spatial_transformer", C.c_int32), ("add_temporal_transformer", C.c_int32),
                ("loc_A", C.c_int32), ("loc_B", C.c_int32), ("loc_C", C.c_int32), ("loc_D", C.c_int32),
                ("loc_M", C.c_int32)]


CB_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p)
CB_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p)
CB_HALO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p)
CB_A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p)


class ClipComm(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
                ("all_gather", CB_GATHER), ("all_reduce_sum_f32", CB_REDUCE), ("halo_exchange", CB_HALO),
                ("user", C.c_void_p), ("ws_needed", C.c_int64), ("all_to_all", CB_A2A), ("next_lane", C.c_void_p)]


# every symbol include/ctrl_hip.h declares (tests check the library exports all of them)
ABI_VERSION = 7       # CTRL_ABI_VERSION of include/ctrl_hip.h
EXPORTS = [
    "ctrl_abi_version", "ctrl_last_error", "ctrl_prof_begin", "ctrl_prof_end", "ctrl_prof_count", "ctrl_prof_get",
    "ctrl_prof_launch_count", "ctrl_prof_launch_get",
    "ctrl_op_igemm", "ctrl_op_ffn", "ctrl_op_ffn_pack_w2", "ctrl_range_check", "ctrl_range_status", "ctrl_igemm_set_order", "ctrl_igemm_set_wide", "ctrl_group_launches", "ctrl_policy_set", "ctrl_policy_get", "ctrl_policy_count", "ctrl_policy_name", "ctrl_igemm_tile_of", "ctrl_op_flash_attn", "ctrl_attn_set_variant", "ctrl_attn_work_map", "ctrl_op_temporal_attn", "ctrl_op_gn_stats_floats", "ctrl_op_gn_stats", "ctrl_op_gn_apply", "ctrl_op_gn_apply_split", "ctrl_op_gn_fused_applies", "ctrl_op_gn_fused", "ctrl_op_pack_conv_w_dup",
    "ctrl_op_layernorm", "ctrl_op_nchw_to_nhwc", "ctrl_op_nhwc_to_nchw", "ctrl_avgpool_nchw",
    "ctrl_op_timestep_sincos", "ctrl_op_linear_small", "ctrl_op_blend", "ctrl_op_add_rowvec",
    "ctrl_op_conv3x3_direct", "ctrl_op_conv3x3_small_mfma", "ctrl_op_pack_conv_w", "ctrl_op_pack_conv_w_direct", "ctrl_op_pack_linear_w",
    "ctrl_op_pack_vec",
    "ctrl_controlnet_param_count", "ctrl_controlnet_param_spec", "ctrl_controlnet_create",
    "ctrl_controlnet_destroy", "ctrl_controlnet_clone", "ctrl_controlnet_forward",
    "ctrl_adapter_param_count", "ctrl_adapter_param_spec", "ctrl_adapter_create", "ctrl_adapter_destroy",
    "ctrl_adapter_forward",
    "ctrl_adapter_forward_scatter", "ctrl_adapter_forward_clip_sharded", "ctrl_controlnet_text_cache", "ctrl_adapter_text_cache", "ctrl_controlnet_trim", "ctrl_adapter_trim", "ctrl_controlnet_selection", "ctrl_adapter_selection",
    "ctrl_step_forward", "ctrl_rccl_unique_id", "ctrl_rccl_comm_create", "ctrl_rccl_comm_destroy", "ctrl_rccl_comm_bind", "ctrl_rccl_comm_bytes_sent",
    "ctrl_router_weights", "ctrl_router_merge", "ctrl_prepare_images",
]

_lib = None


def lib():
    """Loads libctrlhip.so once.  Raises if it is missing -- there is no CPU/eager fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libctrlhip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). The HIP path has no fallback." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.ctrl_last_error.restype = C.c_char_p
        for name in EXPORTS:
            fn = getattr(_lib, name)
            if name not in ("ctrl_last_error", "ctrl_policy_get", "ctrl_policy_name", "ctrl_controlnet_destroy", "ctrl_adapter_destroy", "ctrl_rccl_comm_destroy", "ctrl_rccl_comm_bytes_sent"):
                fn.restype = C.c_int
        _lib.ctrl_policy_get.restype = C.c_char_p
        _lib.ctrl_policy_get.argtypes = [C.c_char_p]
        _lib.ctrl_policy_name.restype = C.c_char_p
        _lib.ctrl_policy_set.argtypes = [C.c_char_p, C.c_char_p]
        _lib.ctrl_controlnet_destroy.restype = None
        _lib.ctrl_adapter_destroy.restype = None
        _lib.ctrl_rccl_comm_destroy.restype = None
        _lib.ctrl_rccl_comm_destroy.argtypes = [C.c_void_p]
        _lib.ctrl_rccl_comm_bytes_sent.restype = C.c_int64
        _lib.ctrl_rccl_comm_bytes_sent.argtypes = [C.c_void_p]
        _lib.ctrl_rccl_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        _lib.ctrl_rccl_comm_bind.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        _lib.ctrl_rccl_unique_id.argtypes = [C.c_void_p]
        _lib.ctrl_op_gn_stats_floats.restype = C.c_size_t
        if _lib.ctrl_abi_version() != ABI_VERSION:
            raise RuntimeError("libctrlhip ABI version mismatch")
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().ctrl_last_error()
        raise RuntimeError("libctrlhip: " + (msg.decode() if msg else "unknown error"))


def range_check(on=None):
    """fp16 range check of the activations (debug aid for real checkpoints): range_check(True / False) switches it, range_check() queries.
    While on (also: CTRL_CHECK_FINITE=1), every forward of the module mirrors ends with range_status() and raises RuntimeError when a
    GEMM / convolution epilogue wrote an fp16 inf / nan (an activation beyond 65504)."""
    return bool(lib().ctrl_range_check(-1 if on is None else (1 if on else 0)))


def raise_if_out_of_range(what):
    """called by the module mirrors after a forward; synchronises only while the check is on, and never during a graph capture"""
    import torch
    if not lib().ctrl_range_check(-1) or torch.cuda.is_current_stream_capturing():
        return
    if lib().ctrl_range_status(1) == 1:
        raise RuntimeError("libctrlhip: %s produced a non-finite fp16 activation (a value beyond the fp16 range, |x| > 65504, or a NaN): "
                           "the fp16-operand path cannot represent this checkpoint's activations" % what)


def dtype_code(t):
    import torch
    if t == torch.float32:
        return F32
    if t == torch.float16:
        return F16
    if t == torch.bfloat16:
        return BF16
    raise ValueError("unsupported dtype %s (float32, float16, bfloat16 are supported)" % t)


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
