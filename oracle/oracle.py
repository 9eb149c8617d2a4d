"""ctypes/numpy front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.

The helpers mirror the reference's Python-level operator surface so parity
tests read like calls into the reference:
  grid_encode_forward  <-> gridencoder/grid.py:24-69   (returns [B, L*C])
  grid_encode_backward <-> gridencoder/grid.py:71-95
  sh_encode / freq_encode / sample_pdf / near_far_from_aabb / contract / render
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

MAX_LEVELS = 32
MAX_LAYERS = 8
MAX_STAGES = 4


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (gcc, OpenMP)."""
    import fcntl
    src = os.path.join(_HERE, "oracle.c")

    def stale():
        return force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "oracle.h")))

    if stale():
        # several processes may get here at once (pytest -n, one rank per GPU): one builds, the others wait and re-check
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if stale():
                    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB_PATH


class OrcMLP(C.Structure):
    _fields_ = [
        ("weight", C.c_void_p * MAX_LAYERS),
        ("bias", C.c_void_p * MAX_LAYERS),
        ("dims", C.c_uint32 * (MAX_LAYERS + 1)),
        ("num_layers", C.c_uint32),
        ("activation", C.c_uint32),
        ("skip_mask", C.c_uint32),
    ]


class OrcGrid(C.Structure):
    _fields_ = [
        ("embeddings", C.c_void_p),
        ("table_dtype", C.c_int),
        ("offsets", C.c_int32 * (MAX_LEVELS + 1)),
        ("D", C.c_uint32),
        ("C", C.c_uint32),
        ("L", C.c_uint32),
        ("S", C.c_float),
        ("H", C.c_uint32),
        ("gridtype", C.c_uint32),
        ("align_corners", C.c_uint32),
        ("interp", C.c_uint32),
    ]


class OrcRenderCfg(C.Structure):
    _fields_ = [
        ("num_stages", C.c_uint32),
        ("num_steps", C.c_uint32 * MAX_STAGES),
        ("prop_grid", OrcGrid * MAX_STAGES),
        ("prop_mlp", OrcMLP * MAX_STAGES),
        ("grid", OrcGrid),
        ("grid_mlp", OrcMLP),
        ("view_mlp", OrcMLP),
        ("sh_degree", C.c_uint32),
        ("aabb", C.c_float * 6),
        ("min_near", C.c_float),
        ("bound", C.c_float),
        ("contract", C.c_int),
        ("last_sample_opaque", C.c_int),
        ("bg_color", C.c_float),
        ("with_sam", C.c_int),
        ("s_grid", OrcGrid),
        ("samvit_mlp", OrcMLP),
        ("ln_weight", C.c_void_p),
        ("ln_bias", C.c_void_p),
        ("ln_eps", C.c_float),
        ("with_mask", C.c_int),
        ("m_grid", OrcGrid),
        ("mask_mlp", OrcMLP),
    ]


class OrcRenderDebug(C.Structure):
    _fields_ = [
        ("nears", C.c_void_p),
        ("fars", C.c_void_p),
        ("bins", C.c_void_p * MAX_STAGES),
        ("real_bins", C.c_void_p * MAX_STAGES),
        ("sigmas", C.c_void_p * MAX_STAGES),
        ("weights", C.c_void_p * MAX_STAGES),
        ("inds", C.c_void_p * MAX_STAGES),
        ("xyzs_last", C.c_void_p),
        ("f_image", C.c_void_p),
        ("u_table", C.c_void_p * MAX_STAGES),
        ("bins0_table", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_expf.restype = C.c_float
        _lib.orc_expf.argtypes = [C.c_float]
        _lib.orc_half_to_float.restype = C.c_float
        _lib.orc_half_to_float.argtypes = [C.c_uint16]
        _lib.orc_float_to_half.restype = C.c_uint16
        _lib.orc_float_to_half.argtypes = [C.c_float]
        _lib.orc_level_resolution.restype = C.c_uint32
        _lib.orc_level_resolution.argtypes = [C.c_uint32, C.c_float, C.c_uint32]
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_grid_row.restype = C.c_uint32
        _lib.orc_grid_row.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]
    return _lib


def grid_row(gridtype: int, hashmap_size: int, resolution: int, pos_grid) -> int:
    """Table row of one grid vertex (gridencoder.cu:45-79)."""
    pg = (C.c_uint32 * len(pos_grid))(*[int(v) & 0xFFFFFFFF for v in pos_grid])
    return int(lib().orc_grid_row(int(gridtype), int(hashmap_size), int(resolution), pg, len(pos_grid)))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------
# grid layout helpers (gridencoder/grid.py:103-142)
# --------------------------------------------------------------------------
def grid_layout(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2.0, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=None):
    """Row offsets + per_level_scale exactly as GridEncoder.__init__ computes them."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    max_params = 2 ** log2_hashmap_size
    offsets, offset = [], 0
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params_in_level = min(max_params, resolution ** input_dim)
        params_in_level = int(np.ceil(params_in_level / 8) * 8)
        offsets.append(offset)
        offset += params_in_level
    offsets.append(offset)
    return np.asarray(offsets, dtype=np.int32), float(per_level_scale)


def level_resolutions(L: int, S: float, H: int) -> List[int]:
    return [int(lib().orc_level_resolution(l, C.c_float(np.float32(S)), H)) for l in range(L)]


def expf(x):
    x = _f32(x)
    out = np.empty_like(x)
    l = lib()
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = l.orc_expf(C.c_float(flat_in[i]))
    return out


def _table(embeddings: np.ndarray):
    if embeddings.dtype == np.float16:
        return np.ascontiguousarray(embeddings), 1
    return _f32(embeddings), 0


def grid_encode_forward(inputs, embeddings, offsets, per_level_scale, base_resolution,
                        calc_grad_inputs=False, gridtype=0, align_corners=False, interpolation=0,
                        max_level=None):
    """inputs [B,D] in [0,1] -> ([B, L*C] float32, dy_dx [B, L*D*C] or None)."""
    inputs = _f32(inputs)
    emb, dt = _table(embeddings)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    Cc = emb.shape[1]
    S = np.float32(np.log2(per_level_scale))
    max_level = L if max_level is None else min(max_level, L)
    out = np.zeros((L, B, Cc), dtype=np.float32)
    dy_dx = np.zeros((B, L * D * Cc), dtype=np.float32) if calc_grad_inputs else None
    lib().orc_grid_encode_forward(_ptr(inputs), _ptr(emb), C.c_int(dt), _ptr(offsets), _ptr(out),
                                  C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L),
                                  C.c_uint32(max_level), C.c_float(S), C.c_uint32(base_resolution),
                                  _ptr(dy_dx), C.c_uint32(gridtype), C.c_int(int(align_corners)),
                                  C.c_uint32(interpolation))
    return np.ascontiguousarray(out.transpose(1, 0, 2)).reshape(B, L * Cc), dy_dx


def grid_encode_backward(grad, inputs, embeddings, offsets, per_level_scale, base_resolution,
                         dy_dx=None, gridtype=0, align_corners=False, interpolation=0, max_level=None):
    """grad [B, L*C] -> (grad_embeddings [rows,C] float32, grad_inputs [B,D] or None)."""
    inputs = _f32(inputs)
    emb, dt = _table(embeddings)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    Cc = emb.shape[1]
    S = np.float32(np.log2(per_level_scale))
    max_level = L if max_level is None else min(max_level, L)
    g = np.ascontiguousarray(_f32(grad).reshape(B, L, Cc).transpose(1, 0, 2))
    ge = np.zeros(emb.shape, dtype=np.float32)
    gi = np.zeros((B, D), dtype=np.float32) if dy_dx is not None else None
    dd = _f32(dy_dx) if dy_dx is not None else None
    lib().orc_grid_encode_backward(_ptr(g), _ptr(inputs), _ptr(emb), C.c_int(dt), _ptr(offsets), _ptr(ge),
                                   C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L),
                                   C.c_uint32(max_level), C.c_float(S), C.c_uint32(base_resolution),
                                   _ptr(dd), _ptr(gi), C.c_uint32(gridtype), C.c_int(int(align_corners)),
                                   C.c_uint32(interpolation))
    return ge, gi


def grad_total_variation(inputs, embeddings, grad, offsets, weight, per_level_scale, base_resolution,
                         gridtype=0, align_corners=False):
    inputs = _f32(inputs); emb = _f32(embeddings)
    assert grad.dtype == np.float32 and grad.flags["C_CONTIGUOUS"]
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    S = np.float32(np.log2(per_level_scale))
    lib().orc_grad_total_variation(_ptr(inputs), _ptr(emb), _ptr(grad), _ptr(offsets), C.c_float(weight),
                                   C.c_uint32(B), C.c_uint32(D), C.c_uint32(emb.shape[1]), C.c_uint32(L),
                                   C.c_float(S), C.c_uint32(base_resolution), C.c_uint32(gridtype),
                                   C.c_int(int(align_corners)))
    return grad


def grad_weight_decay(embeddings, grad, offsets, weight):
    emb = _f32(embeddings)
    assert grad.dtype == np.float32 and grad.flags["C_CONTIGUOUS"]
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    lib().orc_grad_weight_decay(_ptr(emb), _ptr(grad), _ptr(offsets), C.c_float(weight),
                                C.c_uint32(emb.shape[0]), C.c_uint32(emb.shape[1]),
                                C.c_uint32(offsets.shape[0] - 1))
    return grad


def sh_encode_forward(inputs, degree, calc_grad_inputs=False):
    inputs = _f32(inputs)
    B, D = inputs.shape
    out = np.empty((B, degree * degree), dtype=np.float32)
    dy_dx = np.empty((B, D * degree * degree), dtype=np.float32) if calc_grad_inputs else None
    lib().orc_sh_encode_forward(_ptr(inputs), _ptr(out), C.c_uint32(B), C.c_uint32(D), C.c_uint32(degree), _ptr(dy_dx))
    return out, dy_dx


def sh_dy_dx_fd(inputs, degree):
    """dy_dx [B, 3*degree^2] by a 4th-order central difference of the fp64 forward (test helper)."""
    inputs = _f32(inputs)
    B, D = inputs.shape
    dy_dx = np.empty((B, D * degree * degree), dtype=np.float32)
    lib().orc_sh_dy_dx_fd(_ptr(inputs), C.c_uint32(B), C.c_uint32(D), C.c_uint32(degree), _ptr(dy_dx))
    return dy_dx


def sh_encode_backward(grad, inputs, degree, dy_dx):
    grad = _f32(grad); inputs = _f32(inputs); dy_dx = _f32(dy_dx)
    B, D = inputs.shape
    gi = np.zeros((B, D), dtype=np.float32)
    lib().orc_sh_encode_backward(_ptr(grad), _ptr(inputs), C.c_uint32(B), C.c_uint32(D), C.c_uint32(degree), _ptr(dy_dx), _ptr(gi))
    return gi


def freq_encode_forward(inputs, degree):
    inputs = _f32(inputs)
    B, D = inputs.shape
    Cc = D + D * 2 * degree
    out = np.empty((B, Cc), dtype=np.float32)
    lib().orc_freq_encode_forward(_ptr(inputs), C.c_uint32(B), C.c_uint32(D), C.c_uint32(degree), C.c_uint32(Cc), _ptr(out))
    return out


def freq_encode_backward(grad, outputs, input_dim, degree):
    grad = _f32(grad); outputs = _f32(outputs)
    B, Cc = outputs.shape
    gi = np.empty((B, input_dim), dtype=np.float32)
    lib().orc_freq_encode_backward(_ptr(grad), _ptr(outputs), C.c_uint32(B), C.c_uint32(input_dim), C.c_uint32(degree), C.c_uint32(Cc), _ptr(gi))
    return gi


def generate_rays(pose, fx, fy, cx, cy, H, W):
    pose = _f32(pose).reshape(4, 4)
    ro = np.empty((H * W, 3), dtype=np.float32); rd = np.empty((H * W, 3), dtype=np.float32)
    lib().orc_generate_rays(_ptr(pose), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                            C.c_uint32(H), C.c_uint32(W), _ptr(ro), _ptr(rd))
    return ro, rd


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.05):
    ro = _f32(rays_o); rd = _f32(rays_d); ab = _f32(aabb)
    N = ro.shape[0]
    near = np.empty((N, 1), dtype=np.float32); far = np.empty((N, 1), dtype=np.float32)
    lib().orc_near_far_from_aabb(_ptr(ro), _ptr(rd), _ptr(ab), C.c_float(min_near), C.c_uint32(N), _ptr(near), _ptr(far))
    return near, far


def contract(x):
    x = _f32(x)
    shape = x.shape
    flat = np.ascontiguousarray(x.reshape(-1, 3))
    z = np.empty_like(flat)
    lib().orc_contract(_ptr(flat), C.c_uint32(flat.shape[0]), _ptr(z))
    return z.reshape(shape)


def linspace(start, end, steps):
    out = np.empty(steps, dtype=np.float32)
    lib().orc_linspace(C.c_float(np.float32(start)), C.c_float(np.float32(end)), C.c_uint32(steps), _ptr(out))
    return out


def sample_pdf(bins, weights, T, u=None):
    bins = _f32(bins); weights = _f32(weights)
    N, T0 = weights.shape
    out = np.empty((N, T), dtype=np.float32)
    inds = np.empty((N, T), dtype=np.int32)
    ut = _f32(u) if u is not None else None
    lib().orc_sample_pdf(_ptr(bins), _ptr(weights), C.c_uint32(N), C.c_uint32(T0), C.c_uint32(T), _ptr(ut), _ptr(out), _ptr(inds))
    return out, inds


def weights_from_sigma(real_bins, sigmas, last_sample_opaque=True):
    rb = _f32(real_bins); sg = _f32(sigmas)
    N, T = sg.shape
    w = np.empty((N, T), dtype=np.float32)
    lib().orc_weights_from_sigma(_ptr(rb), _ptr(sg), C.c_uint32(N), C.c_uint32(T), C.c_int(int(last_sample_opaque)), _ptr(w))
    return w


# --------------------------------------------------------------------------
# struct builders
# --------------------------------------------------------------------------
class _Keep:
    """Keeps numpy buffers referenced by ctypes structs alive."""

    def __init__(self):
        self.bufs: List[np.ndarray] = []

    def hold(self, a):
        self.bufs.append(a)
        return a


def make_mlp(weights: Sequence[np.ndarray], biases: Optional[Sequence[Optional[np.ndarray]]] = None,
             activation: str = "relu", skip_layers: Sequence[int] = (), keep: Optional[_Keep] = None,
             dim_in: Optional[int] = None) -> OrcMLP:
    """weights[l] has torch nn.Linear layout [out, in]."""
    keep = keep or _Keep()
    m = OrcMLP()
    m.num_layers = len(weights)
    m.activation = 0 if activation == "relu" else 1
    m.skip_mask = 0
    for l in skip_layers:
        m.skip_mask |= 1 << l
    m.dims[0] = weights[0].shape[1] if dim_in is None else dim_in
    for l, w in enumerate(weights):
        w = keep.hold(_f32(w))
        m.weight[l] = w.ctypes.data
        m.dims[l + 1] = w.shape[0]
        b = None if biases is None else biases[l]
        if b is not None:
            b = keep.hold(_f32(b))
            m.bias[l] = b.ctypes.data
        else:
            m.bias[l] = None
    m._keep = keep
    return m


def make_grid(embeddings: np.ndarray, offsets: np.ndarray, per_level_scale: float, base_resolution: int = 16,
              input_dim: int = 3, gridtype: int = 0, align_corners: bool = False, interpolation: int = 0,
              keep: Optional[_Keep] = None) -> OrcGrid:
    keep = keep or _Keep()
    g = OrcGrid()
    emb, dt = _table(embeddings)
    keep.hold(emb)
    g.embeddings = emb.ctypes.data
    g.table_dtype = dt
    L = len(offsets) - 1
    for i, o in enumerate(offsets):
        g.offsets[i] = int(o)
    g.D, g.C, g.L = input_dim, emb.shape[1], L
    g.S = np.float32(np.log2(per_level_scale))
    g.H = base_resolution
    g.gridtype, g.align_corners, g.interp = gridtype, int(align_corners), interpolation
    g._keep = keep
    return g


def render(cfg: OrcRenderCfg, rays_o, rays_d, cam_near_far=None, debug: bool = False,
           u_tables: Optional[Dict[int, np.ndarray]] = None, bins0_table: Optional[np.ndarray] = None):
    """orc_render_rays.  Returns dict with image/depth/weights_sum[/samvit/instance_mask_logits]
    and, with debug=True, per-stage bins/real_bins/sigmas/weights/inds."""
    ro = _f32(rays_o); rd = _f32(rays_d)
    N = ro.shape[0]
    image = np.empty((N, 3), dtype=np.float32)
    depth = np.empty(N, dtype=np.float32)
    wsum = np.empty(N, dtype=np.float32)
    samvit = None
    if cfg.with_sam:
        od = cfg.samvit_mlp.dims[cfg.samvit_mlp.num_layers]
        samvit = np.empty((N, od), dtype=np.float32)
    mask = None
    if cfg.with_mask:
        ni = cfg.mask_mlp.dims[cfg.mask_mlp.num_layers]
        mask = np.empty((N, ni), dtype=np.float32)
    cnf = _f32(cam_near_far) if cam_near_far is not None else None
    if cnf is not None and cnf.shape[0] == 1:
        cnf = np.ascontiguousarray(np.broadcast_to(cnf, (N, 2)))
    dbg = OrcRenderDebug()
    out: Dict[str, np.ndarray] = {}
    held = []
    S = cfg.num_stages
    if u_tables:
        for k, u in u_tables.items():
            u = _f32(u); held.append(u); dbg.u_table[k] = u.ctypes.data
    if bins0_table is not None:
        b0 = _f32(bins0_table); held.append(b0); dbg.bins0_table = b0.ctypes.data
    if debug:
        out["nears"] = np.empty(N, dtype=np.float32); dbg.nears = out["nears"].ctypes.data
        out["fars"] = np.empty(N, dtype=np.float32); dbg.fars = out["fars"].ctypes.data
        for k in range(S):
            T = cfg.num_steps[k]
            for name, shape, dt in (("bins", (N, T + 1), np.float32), ("real_bins", (N, T + 1), np.float32),
                                    ("sigmas", (N, T), np.float32), ("weights", (N, T), np.float32)):
                a = np.empty(shape, dtype=dt)
                out[f"{name}{k}"] = a
                getattr(dbg, name)[k] = a.ctypes.data
            if k >= 1:
                a = np.empty((N, T + 1), dtype=np.int32)
                out[f"inds{k}"] = a
                dbg.inds[k] = a.ctypes.data
        Tl = cfg.num_steps[S - 1]
        out["xyzs_last"] = np.empty((N, Tl, 3), dtype=np.float32); dbg.xyzs_last = out["xyzs_last"].ctypes.data
        ncol = cfg.grid_mlp.dims[cfg.grid_mlp.num_layers] - 1 + cfg.sh_degree ** 2
        out["f_image"] = np.empty((N, ncol), dtype=np.float32); dbg.f_image = out["f_image"].ctypes.data
    lib().orc_render_rays(C.byref(cfg), _ptr(ro), _ptr(rd), C.c_uint32(N), _ptr(cnf),
                          _ptr(image), _ptr(depth), _ptr(wsum), _ptr(samvit), _ptr(mask), C.byref(dbg))
    out.update(image=image, depth=depth, weights_sum=wsum)
    if samvit is not None:
        out["samvit"] = samvit
    if mask is not None:
        out["instance_mask_logits"] = mask
    return out


def mlp_forward(m: OrcMLP, x):
    x = _f32(x)
    B = x.shape[0]
    y = np.empty((B, m.dims[m.num_layers]), dtype=np.float32)
    lib().orc_mlp_forward(C.byref(m), _ptr(x), C.c_uint32(B), _ptr(y))
    return y


def num_threads() -> int:
    return int(lib().orc_num_threads())
