"""ctypes binding of libsanerf_hip.so (the C ABI declared in include/sanerf_hip.h).

This is the only place the product touches native code.  There is no CPU or
eager-torch fallback: if the library is missing, or an operator is handed a
tensor that is not on a HIP device, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SN_LIB") or os.path.join(_PKG, "libsanerf_hip.so")   # SN_LIB: A/B a differently built library
CSRC = os.path.join(_PKG, "csrc")

MAX_LEVELS, MAX_LAYERS, MAX_STAGES = 32, 8, 4
SN_F32, SN_F16 = 0, 1
LAYOUT_LBC, LAYOUT_BLC = 0, 1


class GridDesc(C.Structure):
    _fields_ = [("embeddings", C.c_void_p), ("table_dtype", C.c_int32), ("offsets", C.c_int32 * (MAX_LEVELS + 1)),
                ("D", C.c_uint32), ("C", C.c_uint32), ("L", C.c_uint32), ("S", C.c_float), ("H", C.c_uint32),
                ("gridtype", C.c_uint32), ("align_corners", C.c_uint32), ("interp", C.c_uint32)]


class MlpDesc(C.Structure):
    _fields_ = [("weight", C.c_void_p * MAX_LAYERS), ("bias", C.c_void_p * MAX_LAYERS),
                ("dims", C.c_uint32 * (MAX_LAYERS + 1)), ("num_layers", C.c_uint32),
                ("activation", C.c_uint32), ("skip_mask", C.c_uint32)]


MLP_AUTO, MLP_F16X3, MLP_MFMA32, MLP_VALU, MLP_F16X1 = 0, 1, 2, 3, 5              # sn_render_tuning.mlp_mode
EXP_NONE, EXP_ROLE_SPLIT, EXP_LDS_LEVEL0, EXP_FINAL_ONE_WG = 0, 1, 2, 3                   # sn_render_tuning.experiment (experiments builds only)
BUILD_EXPERIMENTS, BUILD_POISON_LDS = 1, 2                           # sn_build_flags()
ADAM_ZERO_GRAD, ADAM_LAZY = 1, 2                                     # sn_adam_step flags


class RenderTuning(C.Structure):
    _fields_ = [("mlp_mode", C.c_int32), ("per_sample_form", C.c_int32), ("densify", C.c_int32), ("linear_tile_order", C.c_int32),
                ("prop_sp_max_rays", C.c_int32), ("final_sp_max_rays", C.c_int32), ("feat_levels", C.c_int32), ("band_streams", C.c_int32), ("exact_early_out", C.c_int32), ("wave_tile", C.c_int32), ("prop_sp_lanes", C.c_int32), ("feat_patch", C.c_int32), ("prop_pair", C.c_int32), ("experiment", C.c_int32)]


class LaunchInfo(C.Structure):
    _fields_ = [("final_kernel", C.c_char * 64), ("workgroups", C.c_uint32), ("lds_bytes", C.c_uint32), ("dense_levels", C.c_uint32),
                ("gathers_per_wave_sample", C.c_uint32), ("launches", C.c_uint32)]


class RenderCfg(C.Structure):
    _fields_ = [("num_stages", C.c_uint32), ("num_steps", C.c_uint32 * MAX_STAGES),
                ("prop_grid", GridDesc * MAX_STAGES), ("prop_mlp", MlpDesc * MAX_STAGES),
                ("grid", GridDesc), ("grid_mlp", MlpDesc), ("view_mlp", MlpDesc),
                ("sh_degree", C.c_uint32), ("aabb", C.c_float * 6), ("min_near", C.c_float), ("bound", C.c_float),
                ("contract", C.c_int32), ("last_sample_opaque", C.c_int32), ("bg_color", C.c_float),
                ("feat_grid", GridDesc), ("with_feat", C.c_int32), ("early_stop_eps", C.c_float), ("mlp_exact_fp32", C.c_int32),
                ("compact_live", C.c_int32), ("tuning", RenderTuning)]


class RenderIO(C.Structure):
    _fields_ = [("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("cam_near_far", C.c_void_p),
                ("N", C.c_uint32), ("tile_w", C.c_uint32),
                ("bins0_table", C.c_void_p), ("u_table", C.c_void_p * MAX_STAGES),
                ("bins0_ray_stride", C.c_uint32), ("u_ray_stride", C.c_uint32 * MAX_STAGES), ("skip_final", C.c_int32),
                ("image", C.c_void_p), ("depth", C.c_void_p), ("weights_sum", C.c_void_p), ("out_stride", C.c_uint32),
                ("bins", C.c_void_p * MAX_STAGES), ("weights", C.c_void_p * MAX_STAGES),
                ("sigmas", C.c_void_p * MAX_STAGES), ("inds", C.c_void_p * MAX_STAGES),
                ("xyzs_last", C.c_void_p), ("geo_feat_last", C.c_void_p), ("f_image", C.c_void_p), ("f_feat", C.c_void_p), ("head_stride", C.c_uint32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


_u32, _f32, _i32, _vp, _int = C.c_uint32, C.c_float, C.c_int32, C.c_void_p, C.c_int
ABI_VERSION = 12  # include/sanerf_hip.h: SN_ABI_VERSION

_SIGNATURES = {
    "sn_abi_version": (_int, []),
    "sn_build_flags": (_int, []),
    "sn_debug_set": (_int, [C.c_char_p, _int]),
    "sn_last_error": (C.c_char_p, []),
    "sn_device_count": (_int, []),
    "sn_grid_encode_forward": (_int, [_vp, _vp, _int, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _u32, _int, _u32, _int, _vp]),
    "sn_grid_encode_forward_cat": (_int, [_vp, _vp, _int, _vp, _vp, _u32, _vp, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _vp]),
    "sn_grid_encode_backward": (_int, [_vp, _vp, _vp, _int, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32, _int, _u32, _int, _vp]),
    "sn_grid_backward_binned_workspace_bytes": (C.c_size_t, [_u32, _u32, _u32, _u32, _u32, _vp]),
    "sn_grid_encode_backward_binned": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _int, _vp, C.c_size_t, _vp]),
    "sn_grid_encode_backward_binned_rows": (_int, [_vp, _u32, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32, _int, _vp, C.c_size_t, _vp]),
    "sn_grad_total_variation": (_int, [_vp, _vp, _vp, _vp, _f32, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _int, _vp]),
    "sn_grad_weight_decay": (_int, [_vp, _vp, _vp, _f32, _u32, _u32, _u32, _vp]),
    "sn_sh_encode_forward": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp]),
    "sn_sh_encode_backward": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "sn_freq_encode_forward": (_int, [_vp, _u32, _u32, _u32, _u32, _vp, _vp]),
    "sn_freq_encode_backward": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp]),
    "sn_rm_generate_rays": (_int, [_vp, _f32, _f32, _f32, _f32, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "sn_rm_rays_from_pixels": (_int, [_vp, _u32, _vp, _u32, _vp, _u32, _u32, _vp, _vp, _vp]),
    "sn_rm_near_far_from_aabb": (_int, [_vp, _vp, _vp, _f32, _u32, _vp, _vp, _vp]),
    "sn_rm_contract": (_int, [_vp, _u32, _vp, _vp]),
    "sn_rm_sample_pdf": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _u32, _vp, _vp, _vp]),
    "sn_rm_weights_from_sigma": (_int, [_vp, _vp, _u32, _u32, _int, _vp, _vp]),
    "sn_rm_weights_from_sigma_backward": (_int, [_vp, _vp, _vp, _u32, _u32, _int, _vp, _vp]),
    "sn_rm_distort_loss": (_int, [_vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "sn_rm_proposal_loss": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "sn_rm_sample_positions": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _int, _vp, _vp, _vp, _vp]),
    "sn_rm_sample_positions_ex": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _int, _f32, _vp, _vp, _vp, _vp]),
    "sn_rm_jitter": (_int, [_vp, _u32, _u32, _int, _vp, _vp]),
    "sn_rm_ray_composite": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "sn_rm_ray_composite_backward": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "sn_rm_proposal_loss_scaled": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp, _vp, _vp, _vp]),
    "sn_gemm_f32": (_int, [_vp, C.c_int64, C.c_int64, _vp, C.c_int64, C.c_int64, _vp, C.c_int32, _u32, _u32, _u32, _vp, C.c_int64, _vp]),
    "sn_rm_proposal_loss_workspace_bytes": (C.c_size_t, [_u32, _u32, _u32, _int]),
    "sn_rm_proposal_loss_long": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "sn_zero": (_int, [_vp, C.c_size_t, _vp]),
    "sn_mlp_small_supported": (_int, [C.POINTER(MlpDesc)]),
    "sn_mlp_small_forward_train": (_int, [C.POINTER(MlpDesc), _vp, _u32, _vp, _vp, _i32, _vp, _f32, _vp, _vp]),
    "sn_mlp_small_backward": (_int, [C.POINTER(MlpDesc), _vp, _vp, _i32, _vp, _f32, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "sn_rm_composite": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp]),
    "sn_rm_composite_backward": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp]),
    "sn_rm_grid_composite": (_int, [_vp, _vp, _u32, _u32, _f32, C.POINTER(GridDesc), _u32, _vp, _vp]),
    "sn_mlp_wide_workspace_bytes": (C.c_size_t, [C.POINTER(MlpDesc)]),
    "sn_mlp_wide_forward": (_int, [C.POINTER(MlpDesc), _vp, _vp, _f32, _vp, _u32, _vp, _vp, C.c_size_t, _vp]),
    "sn_mlp_wide_overflow": (_int, [_vp]),
    "sn_mlp_wide_forward_train": (_int, [C.POINTER(MlpDesc), _vp, _u32, _vp, _vp, _vp]),
    "sn_mlp_wide_forward_train_f16x3": (_int, [C.POINTER(MlpDesc), _vp, _u32, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "sn_mlp_wide_backward_bits": (_int, [C.POINTER(MlpDesc), _vp, _vp, _u32, _vp, _vp, _vp, C.c_size_t, _vp]),
    "sn_mlp_wide_backward_workspace_bytes": (C.c_size_t, [C.POINTER(MlpDesc)]),
    "sn_mlp_wide_backward": (_int, [C.POINTER(MlpDesc), _vp, _vp, _u32, _vp, _vp, _vp, C.c_size_t, _vp]),
    "sn_rm_mask_head_workspace_bytes": (C.c_size_t, [C.POINTER(MlpDesc)]),
    "sn_rm_mask_head": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _f32, C.POINTER(GridDesc), C.POINTER(MlpDesc), _vp, _vp, C.c_size_t, _vp]),
    "sn_adam_step": (_int, [_vp, _vp, _vp, _vp, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _u32, _vp, _int, _int, _vp]),
    "sn_linear_wgrad_workspace_bytes": (C.c_size_t, [_u32, _u32, _u32]),
    "sn_linear_wgrad": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, C.c_size_t, _vp]),
    "sn_rm_render_workspace_bytes": (C.c_size_t, [C.POINTER(RenderCfg), _u32, _u32]),
    "sn_rm_render_rays": (_int, [C.POINTER(RenderCfg), C.POINTER(RenderIO), _vp]),
    "sn_rm_profile_enable": (None, [_int]),
    "sn_rm_profile_read": (_int, [_vp, _vp, _int]),
    "sn_rm_profile_shader_clock": (_int, [_vp, _vp]),
    "sn_rm_mask_nll": (_int, [_vp, _vp, _u32, _u32, _f32, _vp, _vp, _vp]),
    "sn_rm_debug_occupancy": (_int, [_vp, _vp, _int]),
    "sn_rm_last_launch_info": (_int, [_vp]),
    "sn_debug_eval": (_int, [_int, _vp, _vp, _u32, _vp, _vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "sanerf_hip.h"))
    stale = force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", CSRC, "-j4"] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    """The loaded library; raises if it was never built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU / eager fallback for these operators.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                if os.environ.get("SN_LIB"):        # A/B against an older build of the same ABI version: newer entry points are simply absent
                    continue
                raise ImportError(f"{LIB_PATH} does not export {name}: rebuild it")
            fn.restype, fn.argtypes = res, args
        if l.sn_abi_version() != ABI_VERSION:   # the ctypes structs above mirror include/sanerf_hip.h of exactly this version
            raise ImportError(f"{LIB_PATH} has ABI version {l.sn_abi_version()}, this package expects {ABI_VERSION}: rebuild it")
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().sn_last_error().decode()
        raise RuntimeError(f"{what}: {msg}" if what else msg)


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def dev(t: Optional[torch.Tensor], name: str, dtype=torch.float32):
    """Device pointer of a contiguous tensor on the HIP device (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")          # gridencoder.cu:15
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")    # gridencoder.cu:16
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must be a {dtype} tensor, got {t.dtype}")
    return t.data_ptr()


def host_i32(values):
    arr = (C.c_int32 * len(values))(*[int(v) for v in values])
    return arr


def host_f32(values):
    return (C.c_float * len(values))(*[float(v) for v in values])
