"""ctypes binding of libkdip_hip.so (include/kdip.h).  Fails loudly if the library is
missing or a compute entry point is called without a GPU -- there is no fallback path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KDIP_LIB_PATH", os.path.join(_HERE, "libkdip_hip.so"))   # override: kernel-variant A/B runs

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_long_p = C.POINTER(C.c_long)
VP = C.c_void_p

# name -> (restype, argtypes); must list every symbol declared in include/kdip.h
SIGNATURES = {
    "kdip_last_error": (C.c_char_p, []),
    "kdip_version": (C.c_int, []),
    "kdip_unet_create": (C.c_int, [C.c_int] * 7 + [c_int_p, C.c_int, c_int_p, C.c_int, C.c_int, C.POINTER(VP)]),
    "kdip_unet_destroy": (None, [VP]),
    "kdip_unet_load": (C.c_int, [VP, C.c_char_p, VP, c_long_p, C.c_int]),
    "kdip_unet_finalize": (C.c_int, [VP]),
    "kdip_unet_forward": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_float, VP, VP, VP]),
    "kdip_unet_vjp": (C.c_int, [VP, VP, VP, C.c_int, VP]),
    "kdip_unet_debug_stash_checksum": (C.c_int, [VP, VP, VP]),
    "kdip_unet_workspace_bytes": (C.c_long, [VP, C.c_int]),
    "kdip_unet_workspace_generation": (C.c_long, [VP]),
    "kdip_unet_x3_window": (C.c_int, [VP, C.c_int]),
    "kdip_unet_deterministic": (C.c_int, [VP, C.c_int]),
    "kdip_unet_x3_saturated": (C.c_int, [VP, VP, C.c_int, C.POINTER(C.c_int)]),
    "kdip_unet_x3_head": (C.c_int, [VP, C.c_int]),
    "kdip_op_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(VP)]),
    "kdip_op_destroy": (None, [VP]),
    "kdip_op_set_psf": (C.c_int, [VP, VP, C.c_int, C.c_int]),
    "kdip_op_set_separable": (C.c_int, [VP, VP, VP, C.c_int]),
    "kdip_op_set_mask": (C.c_int, [VP, VP]),
    "kdip_op_set_ortho": (C.c_int, [VP, C.c_int]),
    "kdip_op_get_otf": (C.c_int, [VP, VP, VP]),
    "kdip_op_apply": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, VP]),
    "kdip_op_solve": (C.c_int, [VP, VP, VP, VP, C.c_float, VP, C.c_int, VP, VP, VP]),
    "kdip_op_ortho": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, VP]),
    "kdip_op_set_cg_fixed_trips": (C.c_int, [VP, C.c_int]),
    "kdip_op_cg_unconverged": (C.c_int, [VP, VP, c_int_p]),
    "kdip_stream_create_cu_mask": (C.c_int, [C.c_int, VP, C.c_int, C.POINTER(VP)]),
    "kdip_stream_destroy": (C.c_int, [VP]),
    "kdip_debug_cu_census": (C.c_int, [VP, C.c_int, VP]),
    "kdip_guided_ws_floats": (C.c_long, [C.c_int, C.c_int]),
    "kdip_guided_ws_layout": (C.c_int, [C.c_int, C.c_int, VP, C.c_int]),
    "kdip_op_workspace_generation": (C.c_long, [VP]),
    "kdip_guided_call_v1": (C.c_int, [VP, VP, VP, VP, VP, VP, C.c_int, VP, C.c_float, C.c_float, C.c_int, VP, VP, VP, VP]),
    "kdip_gather": (C.c_int, [VP, VP, VP, C.c_long, C.c_long, C.c_int, VP]),
    "kdip_scatter": (C.c_int, [VP, VP, VP, C.c_long, C.c_long, C.c_int, VP]),
    "kdip_mask_mul": (C.c_int, [VP, VP, VP, C.c_int, C.c_long, VP]),
    "kdip_resize_axis": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, VP]),
    "kdip_blur_dense": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_long, C.c_int, VP]),
    "kdip_fft2": (C.c_int, [VP, C.c_int, VP, C.c_int, VP, C.c_int, C.c_long, C.c_int, VP]),
    "kdip_x0_epilogue_v1": (C.c_int, [VP, VP, VP, C.c_int, C.c_long, VP, VP, VP, VP]),
    "kdip_x0_epilogue_v2": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_long, C.c_float, C.c_int, VP, VP, VP]),
    "kdip_vjp_cotangent_v1": (C.c_int, [VP, VP, VP, C.c_int, C.c_long, C.c_float, VP, VP]),
    "kdip_vjp_cotangent_v2": (C.c_int, [VP, VP, C.c_int, C.c_long, VP]),
    "kdip_guidance_combine": (C.c_int, [VP, VP, VP, C.c_float, VP, C.c_float, C.c_float, C.c_long, VP]),
    "kdip_axpby": (C.c_int, [VP, VP, C.c_float, VP, C.c_float, C.c_long, VP]),
    "kdip_mul": (C.c_int, [VP, VP, VP, C.c_long, VP]),
    "kdip_clamp": (C.c_int, [VP, VP, C.c_long, VP]),
    "kdip_dps_normalize": (C.c_int, [VP, VP, C.c_long, VP, C.c_long, C.c_float, C.c_int, VP, VP, VP]),
    "kdip_sampler_add_noise": (C.c_int, [VP, VP, VP, C.c_float, C.c_long, VP]),
    "kdip_sampler_euler": (C.c_int, [VP, VP, VP, C.c_float, C.c_float, C.c_long, VP]),
    "kdip_sampler_heun": (C.c_int, [VP, VP, VP, VP, VP, C.c_float, C.c_float, C.c_float, C.c_long, VP]),
    "kdip_gauss_nll_mean": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_long, C.c_int, VP]),
    "kdip_conv_create": (C.c_int, [C.c_int, C.c_int, VP, VP, C.c_int, C.c_int, C.c_int, C.POINTER(VP)]),
    "kdip_conv_destroy": (None, [VP]),
    "kdip_conv_workspace_bytes": (C.c_long, [VP, C.c_int, C.c_int, C.c_int]),
    "kdip_conv_apply": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_int, VP, VP]),
    "kdip_relu_maxpool": (C.c_int, [VP, VP, C.c_long, C.c_int, C.c_int, C.c_int, VP]),
    "kdip_lpips_layer": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_int, C.c_long, VP]),
    "kdip_profile_enable": (C.c_int, [C.c_int]),
    "kdip_profile_num_classes": (C.c_int, []),
    "kdip_profile_class_name": (C.c_char_p, [C.c_int]),
    "kdip_profile_report": (C.c_int, [VP, VP, VP, VP]),
    "kdip_profile_dump": (C.c_int, [C.c_char_p]),
    "kdip_debug_conv_timing": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int]),
    "kdip_test_conv": (C.c_int, [VP, C.c_int, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, C.c_int, C.c_int, VP, C.c_int]),
    "kdip_debug_conv3_timing": (C.c_int, [VP]),
    "kdip_debug_gn_fold": (C.c_int, [C.c_int]),
    "kdip_debug_defer_finish": (C.c_int, [C.c_int]),
    "kdip_debug_x3_peaks": (C.c_int, [VP, VP, VP, C.c_int, C.POINTER(C.c_int)]),
    "kdip_test_attention": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_int, VP, VP]),
    "kdip_test_conv3": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, C.c_int, C.c_int, C.c_int, VP, VP, C.c_int, C.c_int,
                                  C.c_int, VP, VP, VP, VP, VP, C.c_int, VP]),
    "kdip_test_groupnorm": (C.c_int, [VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, VP, VP]),
}

F32, BF16, BF16X3, F16X3 = 0, 1, 2, 3
DTYPES = {"f32": F32, "bf16": BF16, "bf16x3": BF16X3, "f16x3": F16X3}      # include/kdip.h: KDIP_F32 / KDIP_BF16 / KDIP_BF16X3 / KDIP_F16X3
OP_INPAINT, OP_BLUR, OP_SR = 0, 1, 2
GWS_OUT6, GWS_X0_MEAN, GWS_X0_RAW, GWS_VAR, GWS_MAT, GWS_COT, GWS_G_RAW, GWS_UG, GWS_SCORE, GWS_COUNT = range(10)      # include/kdip.h KDIP_GWS_*
OT_NONE, OT_DWT, OT_DCT = 0, 1, 2

_lib = None


class KdipError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed to load it or to resolve symbols)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python k-diffusion-inverse-problems_amd/build.py` "
            "(or __graft_entry__.build()).  kdip_amd has no CPU / PyTorch fallback.")
    # PyTorch-ROCm bundles its own libamdhip64: it has to be in the process BEFORE libkdip_hip.so is loaded, so that the library's
    # libamdhip64.so.7 dependency resolves to the runtime torch allocates with.  Loaded first, the library binds the system runtime
    # instead and the process ends up with two HIP runtimes (the second one reports "no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().kdip_last_error()
        raise KdipError(f"libkdip_hip error {rc}: {msg.decode() if msg else '?'}")


def ptr(t):
    """device/host pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "kdip_amd: tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


def stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise KdipError("kdip_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
