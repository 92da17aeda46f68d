"""Loader of libdimo_hip.so (the C-ABI HIP library) through ctypes.

The product path has NO fallback: if the library is missing or cannot be loaded,
every op raises.  (The CPU oracle under oracle/ is test infrastructure and is
never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdimo_hip.so")
_lib = None

c_f32p = C.c_void_p  # device pointers are passed as integers
c_ptr = C.c_void_p

# name -> (restype, argtypes).  Mirrors include/dimo_hip.h one to one.
_SIGNATURES = {
    "dimo_version": (C.c_char_p, []),
    "dimo_last_error": (C.c_char_p, []),
    "dimo_timing_enable": (C.c_int, [C.c_int]),
    "dimo_timing_select": (C.c_int, [C.c_char_p]),
    "dimo_timing_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dimo_raster_geom_bytes": (C.c_size_t, [C.c_int]),
    "dimo_raster_bin_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    "dimo_raster_img_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dimo_raster_geom_layout": (C.c_int, [C.c_int, C.POINTER(C.c_size_t)]),
    "dimo_raster_bin_layout": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "dimo_raster_img_layout": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "dimo_debug_bin_trace": (C.c_int64, [C.c_void_p, C.c_int64]),
    "dimo_raster_depth_keys": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dimo_debug_bin_geom_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "dimo_debug_bin_instances": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dimo_raster_preprocess_forward": (C.c_int, [C.c_int] * 5 + [c_ptr] * 7 + [C.c_float] + [c_ptr] * 3
                                       + [C.c_float, C.c_float, c_ptr, c_ptr, C.c_size_t, C.POINTER(C.c_int64), c_ptr]),
    "dimo_raster_render_forward": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, c_ptr, c_ptr, c_ptr, C.c_size_t,
                                             c_ptr, C.c_size_t, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dimo_raster_backward_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int64]),
    "dimo_raster_backward": (C.c_int, [C.c_int] * 5 + [C.c_int64] + [c_ptr] * 7 + [C.c_float] + [c_ptr] * 4
                             + [C.c_float, C.c_float] + [c_ptr] * 4 + [c_ptr] * 4 + [c_ptr] * 8
                             + [c_ptr, C.c_size_t, c_ptr]),
    "dimo_knn": (C.c_int, [C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dimo_knn_seeded": (C.c_int, [C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dimo_dist2": (C.c_int, [C.c_int, c_ptr, c_ptr, c_ptr]),
    "dimo_dist2_workspace_bytes": (C.c_size_t, [C.c_int]),
    "dimo_dist2_grid": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dimo_deform_max_ctrl_points": (C.c_int, []),
    "dimo_deform_backward_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dimo_deform_forward": (C.c_int, [C.c_int] * 3 + [c_ptr] * 15),
    "dimo_deform_backward": (C.c_int, [C.c_int] * 4 + [c_ptr] * 22 + [c_ptr, C.c_size_t, c_ptr]),
    "dimo_ssim_forward": (C.c_int, [C.c_int] * 5 + [c_ptr] * 5),
    "dimo_ssim_backward": (C.c_int, [C.c_int] * 5 + [c_ptr] * 6),
    "dimo_ssim_forward_backward": (C.c_int, [C.c_int] * 5 + [c_ptr] * 6),
    "dimo_ssim_forward_backward_images": (C.c_int, [C.c_int] * 5 + [c_ptr, C.POINTER(C.c_void_p)] + [c_ptr] * 4),
    "dimo_timenet_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "dimo_timenet_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_ptr, C.c_void_p, c_ptr, C.c_void_p, c_ptr, c_ptr,
                                       c_ptr, C.c_size_t, c_ptr]),
    "dimo_timenet_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_ptr, c_ptr, C.c_void_p, C.c_void_p, c_ptr,
                                        c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "dimo_farthest_point_sample": (C.c_int, [C.c_int, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "dimo_executor_create": (C.c_void_p, [C.c_int]),
    "dimo_executor_destroy": (None, [C.c_void_p]),
    "dimo_executor_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "dimo_executor_forward_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dimo_executor_join": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dimo_executor_backward_launch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dimo_executor_range_stream": (C.c_void_p, [C.c_void_p, C.c_int]),
    "dimo_executor_backward_launch_in_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                         C.c_void_p]),
    "dimo_executor_side_stream": (C.c_void_p, [C.c_void_p, C.c_int, C.c_void_p]),
    "dimo_executor_side_done": (C.c_int, [C.c_void_p, C.c_int]),
    "dimo_executor_wait_side": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "dimo_executor_private_stream": (C.c_void_p, [C.c_void_p, C.c_int]),
    "dimo_executor_join_ranges": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dimo_executor_backward_skinning_in_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                           C.c_void_p]),
    "dimo_executor_backward_launch_joint": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dimo_executor_backward_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                    C.c_void_p]),
    "dimo_flat_adam_step": (C.c_int, [C.c_int64] + [c_ptr] * 4 + [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_float)]
                            + [C.c_float] * 3 + [C.c_int64, c_ptr, C.c_int, C.c_int, C.c_int, c_ptr]
                            + [c_ptr, C.c_int, c_ptr, C.c_uint32, c_ptr, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                               c_ptr]),
    "dimo_selftest_wave_reduce16": (C.c_int, [c_ptr, c_ptr, c_ptr]),
    "dimo_debug_blend_trace": (C.c_int64, [c_ptr, C.c_int64]),
    "dimo_image_loss": (C.c_int, [C.c_int] * 3 + [c_ptr] * 6 + [C.c_int, C.POINTER(C.c_float)] + [C.c_float] * 5
                        + [c_ptr] * 7 + [C.POINTER(C.c_void_p)] * 2 + [c_ptr]),
    "dimo_ssim_image_loss": (C.c_int, [C.c_int] * 3 + [c_ptr] * 6 + [C.c_int, C.POINTER(C.c_float)] + [C.c_float] * 5
                             + [c_ptr] * 8 + [C.POINTER(C.c_void_p)] * 2 + [c_ptr]),
}

ERRORS = {-1: "DIMO_E_ARG (bad argument)", -2: "DIMO_E_LAUNCH (HIP launch/runtime error)",
          -3: "DIMO_E_WORKSPACE (workspace too small)"}


def exported_symbols():
    """Every symbol include/dimo_hip.h declares (used by the no-GPU export test)."""
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m dimo_amd.csrc.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback in the product path.")
        # torch first: its bundled libamdhip64 must be THE HIP runtime of the process; loading ours before
        # it would pull /opt/rocm's copy in as a second runtime (observed: hipErrorNoDevice on every launch)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        detail = ""
        if rc == -2 and _lib is not None:
            detail = ": " + _lib.dimo_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)}{detail}")


def ptr(t):
    """Device pointer of a (contiguous) tensor or None."""
    return None if t is None else t.data_ptr()


def ptr_array(tensors):
    """HOST array of the device pointers of a list of contiguous fp32 tensors (for the *_images_host arguments)."""
    for t in tensors:
        if not (t.is_cuda and t.is_contiguous() and t.dtype.is_floating_point and t.element_size() == 4):
            raise ValueError("contiguous fp32 GPU tensors expected")
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


_raw_stream = None


def current_stream():
    """Raw handle of torch's current stream on the current device.  (`torch.cuda.current_stream().cuda_stream` builds
    a Stream object per call: ~15 us, a dozen times per step; the raw query is ~0.3 us.)"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
