"""`raymarching` operators on MI355X.

The reference's README (README.md:32-34) mentions a `raymarching` extension that is not in
its tree; the ray-marching maths lives as torch code in nerf/utils.py and nerf/renderer.py.
This module is that operator surface, implemented over libsanerf_hip.so:

    generate_rays        nerf/utils.py:182-304 (full image branch)
    near_far_from_aabb   nerf/renderer.py:122-139
    contract             nerf/renderer.py:60-69
    sample_pdf           nerf/renderer.py:84-119   (+ the integer searchsorted result)
    weights_from_sigma   nerf/renderer.py:308-325
    composite            nerf/renderer.py:333-338, 361, 384  (autograd w.r.t. both operands)
    grid_composite       nerf/renderer.py:301-302 + 361: composite(weights, s_grid(xyzs)) fused (inference)
    mlp_forward          nerf/network.py:31-66 (+ LayerNorm :115): the 256-wide head MLPs on the matrix cores (inference)
    render_rays          nerf/renderer.py:221-357 + nerf/network.py:146-186, fused
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch.autograd import Function

from .. import _lib


class Tuning:
    """Kernel selection of the fused render (sn_render_tuning in include/sanerf_hip.h): plain attributes, copied into the call's
    config right before every launch.  `raymarching.tuning` is the process default; `RenderPlan(..., tuning=Tuning(...))` /
    `render_rays(..., tuning=...)` override it per plan / per call (two threads or two plans may differ -- rounds 1-3 read
    environment variables inside the library instead).  Every field 0 / False = the shipped default; none changes WHAT is
    computed beyond fp32 round-off.
        mlp_mode           _lib.MLP_AUTO | MLP_F16X3 (forces split-fp16 past the range guard) | MLP_MFMA32 (exact fp32) | MLP_VALU
        per_sample_form    the last stage evaluates the third MLP layer per sample everywhere (bit-identical to the compacting and
                           several-lanes-per-ray kernels; the default "linear tail" differs by fp32 round-off)
        densify            0 automatic, 1 never, 2 whenever the kernel exists (levels 5-6 of the main grid re-laid out per call)
        linear_tile_order  no XCD-aware workgroup -> tile remap
        prop_sp_max_rays / final_sp_max_rays   ray-count thresholds of the several-lanes-per-ray kernels (0 default, < 0 never)
        feat_levels        levels per pass of the feature stage (0 default)
        wave_tile          image mode: a wave covers 2^w x 2^(6-w) pixels (0 default = 8x8; 1..5)
        prop_sp_lanes      small linear-order batches: lanes sharing a ray in the proposal stages (0 automatic; 8 / 16 / 32; bit-neutral)
        feat_patch         feature stage: 1 = per-wave LDS patch of the dense levels (bit-neutral; measured 3-9 % slower, so 0 = off is the default)
        prop_pair          proposal stages: a lane evaluates two consecutive samples at once (bit-neutral): 0 automatic, 1 never, 2 always
        experiment         _lib.EXP_*: measured-and-rejected variants, experiments builds only (SN_LIB=.../libsanerf_hip_exp.so)"""
    FIELDS = ("mlp_mode", "per_sample_form", "densify", "linear_tile_order", "prop_sp_max_rays", "final_sp_max_rays", "feat_levels", "band_streams", "exact_early_out", "wave_tile", "prop_sp_lanes", "feat_patch", "prop_pair", "experiment")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, 0)
        for k, v in kw.items():
            if k not in self.FIELDS:
                raise TypeError(f"Tuning: unknown field {k!r} (fields: {', '.join(self.FIELDS)})")
            setattr(self, k, int(v))

    def write(self, ct) -> None:
        for f in self.FIELDS:
            setattr(ct, f, int(getattr(self, f)))

    def __repr__(self):
        return "Tuning(" + ", ".join(f"{f}={getattr(self, f)}" for f in self.FIELDS if getattr(self, f)) + ")"


tuning = Tuning()          # process default; tests and A/B tools set attributes on it (monkeypatch.setattr(rm.tuning, "per_sample_form", 1))
check_range_default = False   # mlp_forward(check_range=None): read the overflow flag after every call (one device sync each)


def last_launch_info() -> dict:
    """What the last render_rays call of this thread launched as its last stage (kernel variant, workgroups, LDS bytes, gather
    instructions per wave-sample): sn_rm_last_launch_info."""
    info = _lib.LaunchInfo()
    _lib.check(_lib.lib().sn_rm_last_launch_info(C.byref(info)), "last_launch_info")
    return dict(final_kernel=info.final_kernel.decode(), workgroups=int(info.workgroups), lds_bytes=int(info.lds_bytes),
                dense_levels=int(info.dense_levels), gathers_per_wave_sample=int(info.gathers_per_wave_sample), launches=int(info.launches))


def _flat3(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(-1, 3).contiguous().float()


def generate_rays(pose, intrinsics, H: int, W: int, device="cuda", row_begin: int = 0, row_end: Optional[int] = None):
    """Rays of image rows [row_begin, row_end): rays_o, rays_d of shape [(rows)*W, 3].
    pose: 4x4 cam2world (array-like / tensor), intrinsics: (fx, fy, cx, cy)."""
    row_end = H if row_end is None else row_end
    pose = np.asarray(pose.detach().cpu() if torch.is_tensor(pose) else pose, dtype=np.float32).reshape(4, 4)
    fx, fy, cx, cy = [float(v) for v in (intrinsics.tolist() if hasattr(intrinsics, "tolist") else intrinsics)]
    n = (row_end - row_begin) * W
    rays_o = torch.empty(n, 3, device=device, dtype=torch.float32)
    rays_d = torch.empty(n, 3, device=device, dtype=torch.float32)
    _lib.check(_lib.lib().sn_rm_generate_rays(_lib.host_f32(pose.reshape(-1)), fx, fy, cx, cy, H, W, row_begin, row_end,
                                              _lib.dev(rays_o, "rays_o"), _lib.dev(rays_d, "rays_d"), _lib.stream()),
               "generate_rays")
    return rays_o, rays_d


def rays_from_pixels(poses: torch.Tensor, intrinsics: torch.Tensor, inds: torch.Tensor, W: int):
    """Rays through the flat pixel indices `inds` [N] (row-major, width W): poses [1 or N, 4, 4] and intrinsics
    [1 or N, 4] on the device, one camera for all rays or one per ray (nerf/utils.py:209-287 with the per-ray cameras of
    provider.py:908-913).  Everything stays on the device."""
    inds = inds.reshape(-1).contiguous().long()
    N = inds.shape[0]
    poses = poses.reshape(-1, 16).contiguous().float()
    intrinsics = intrinsics.reshape(-1, 4).contiguous().float()
    rays_o = torch.empty(N, 3, device=inds.device, dtype=torch.float32)
    rays_d = torch.empty(N, 3, device=inds.device, dtype=torch.float32)
    _lib.check(_lib.lib().sn_rm_rays_from_pixels(_lib.dev(poses, "poses"), poses.shape[0], _lib.dev(intrinsics, "intrinsics"),
                                                 intrinsics.shape[0], _lib.dev(inds, "inds", torch.int64), int(W), N,
                                                 _lib.dev(rays_o, "rays_o"), _lib.dev(rays_d, "rays_d"), _lib.stream()),
               "rays_from_pixels")
    return rays_o, rays_d


def _host_values(t) -> list:
    """Host copy of a small device tensor, memoised on the tensor object (keyed by its in-place version counter): a
    device->host read per call would put a synchronisation into every training step and forbids graph capture."""
    if not torch.is_tensor(t):
        return list(t)
    cached = getattr(t, "_sn_host_values", None)
    if cached is None or cached[0] != t._version:
        cached = (t._version, t.detach().cpu().tolist())
        try:
            t._sn_host_values = cached
        except AttributeError:
            pass
    return cached[1]


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.05):
    rays_o, rays_d = _flat3(rays_o), _flat3(rays_d)
    N = rays_o.shape[0]
    nears = torch.empty(N, 1, device=rays_o.device, dtype=torch.float32)
    fars = torch.empty(N, 1, device=rays_o.device, dtype=torch.float32)
    ab = _host_values(aabb)
    _lib.check(_lib.lib().sn_rm_near_far_from_aabb(_lib.dev(rays_o, "rays_o"), _lib.dev(rays_d, "rays_d"), _lib.host_f32(ab),
                                                   float(min_near), N, _lib.dev(nears, "nears"), _lib.dev(fars, "fars"),
                                                   _lib.stream()), "near_far_from_aabb")
    return nears, fars


def contract(x):
    shape = x.shape
    flat = _flat3(x)
    z = torch.empty_like(flat)
    _lib.check(_lib.lib().sn_rm_contract(_lib.dev(flat, "x"), flat.shape[0], _lib.dev(z, "z"), _lib.stream()), "contract")
    return z.view(shape)


def sample_pdf(bins, weights, T: int, perturb: bool = False, return_inds: bool = False, u: Optional[torch.Tensor] = None):
    """bins [N,T0+1], weights [N,T0] -> [N,T] (no gradient, like the `.detach()` at renderer.py:274)."""
    bins = bins.detach().contiguous().float()
    weights = weights.detach().contiguous().float()
    N, T0 = weights.shape
    out = torch.empty(N, T, device=bins.device, dtype=torch.float32)
    inds = torch.empty(N, T, device=bins.device, dtype=torch.int32) if return_inds else None
    u_stride = 0
    if perturb:   # renderer.py:98-102
        base = torch.linspace(0.5 / T, 1 - 0.5 / T, steps=T, device=bins.device)
        u = (base.expand(N, T) + (torch.rand(N, T, device=bins.device) - 0.5) / T).contiguous()
    if u is not None:
        u = u.contiguous().float()
        u_stride = 0 if u.dim() == 1 else T
    _lib.check(_lib.lib().sn_rm_sample_pdf(_lib.dev(bins, "bins"), _lib.dev(weights, "weights"), N, T0, T,
                                           _lib.dev(u, "u"), u_stride, _lib.dev(out, "out_bins"),
                                           _lib.dev(inds, "inds", torch.int32), _lib.stream()), "sample_pdf")
    return (out, inds) if return_inds else out


class _weights_from_sigma(Function):
    """weights [N,T] of renderer.py:308-325 from real_bins [N,T+1] and sigmas [N,T]; differentiable w.r.t. sigmas (the
    bin edges carry no gradient on this path: sample_pdf's output is not differentiated)."""

    @staticmethod
    def forward(ctx, real_bins, sigmas, last_sample_opaque):
        real_bins = real_bins.detach().contiguous().float()
        sig = sigmas.detach().contiguous().float()
        N, T = sig.shape
        w = torch.empty(N, T, device=sig.device, dtype=torch.float32)
        _lib.check(_lib.lib().sn_rm_weights_from_sigma(_lib.dev(real_bins, "real_bins"), _lib.dev(sig, "sigmas"), N, T,
                                                       int(last_sample_opaque), _lib.dev(w, "weights"), _lib.stream()),
                   "weights_from_sigma")
        ctx.save_for_backward(real_bins, sig)
        ctx.last = int(last_sample_opaque)
        return w

    @staticmethod
    def backward(ctx, gw):
        real_bins, sig = ctx.saved_tensors
        N, T = sig.shape
        gw = gw.contiguous().float()
        gs = torch.empty_like(sig)
        _lib.check(_lib.lib().sn_rm_weights_from_sigma_backward(_lib.dev(real_bins, "real_bins"), _lib.dev(sig, "sigmas"),
                                                                _lib.dev(gw, "grad_weights"), N, T, ctx.last,
                                                                _lib.dev(gs, "grad_sigmas"), _lib.stream()), "weights_from_sigma_backward")
        return None, gs, None


WEIGHTS_BACKWARD_MAX_T = 131072   # sn_rm_weights_from_sigma_backward: any ray the product meets (up to 256 samples in one wave's registers, beyond that the
                                  # segment prefixes in LDS); the torch statement below is what the tests compare the kernel with (they lower this constant)


def weights_from_sigma(real_bins, sigmas, last_sample_opaque: bool = True):
    """real_bins [N,T+1], sigmas [N,T] -> weights [N,T]; under autograd the gradient reaches `sigmas`."""
    if torch.is_grad_enabled() and sigmas.requires_grad:
        if sigmas.shape[-1] > WEIGHTS_BACKWARD_MAX_T:       # (renderer.py:308-325 as torch states it: the tests' comparison route)
            ds = (real_bins[..., 1:] - real_bins[..., :-1]).detach() * sigmas
            if last_sample_opaque:
                ds = torch.cat([ds[..., :-1], torch.full_like(ds[..., -1:], torch.inf)], dim=-1)
            trans = torch.cumsum(ds[..., :-1], dim=-1)
            trans = torch.exp(-torch.cat([torch.zeros_like(ds[..., :1]), trans], dim=-1))
            return ((1 - torch.exp(-ds)) * trans).nan_to_num(0)
        return _weights_from_sigma.apply(real_bins, sigmas, bool(last_sample_opaque))
    return _weights_from_sigma.forward(_NoCtx(), real_bins, sigmas, bool(last_sample_opaque))


class _NoCtx:
    def save_for_backward(self, *a):
        pass


def sample_positions(rays_o, rays_d, nears, fars, bins, contract: bool = True, grid_bound: float = 0.0):
    """One stage's geometry (renderer.py:277-285), no autograd: bins [N,T+1] -> (real_bins [N,T+1], rays_t [N,T],
    xyzs [N,T,3]); positions are contracted into [-2,2]^3 when `contract`.  grid_bound > 0: the positions come back as the
    grid encoder's unit-cube coordinates (x + grid_bound) / (2 grid_bound) (gridencoder/grid.py:156) -- for `grid_encode` itself."""
    rays_o, rays_d = rays_o.detach().contiguous().float(), rays_d.detach().contiguous().float()
    bins = bins.detach().contiguous().float()
    N, T = bins.shape[0], bins.shape[1] - 1
    nears = nears.detach().reshape(-1).contiguous().float()
    fars = fars.detach().reshape(-1).contiguous().float()
    assert nears.numel() == N and fars.numel() == N and rays_o.shape == (N, 3)
    dev = bins.device
    real_bins = torch.empty(N, T + 1, device=dev, dtype=torch.float32)
    rays_t = torch.empty(N, T, device=dev, dtype=torch.float32)
    xyzs = torch.empty(N, T, 3, device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().sn_rm_sample_positions_ex(_lib.dev(rays_o, "rays_o"), _lib.dev(rays_d, "rays_d"), _lib.dev(nears, "nears"),
                                                    _lib.dev(fars, "fars"), _lib.dev(bins, "bins"), N, T, int(contract), float(grid_bound),
                                                    _lib.dev(real_bins, "real_bins"), _lib.dev(rays_t, "rays_t"), _lib.dev(xyzs, "xyzs"),
                                                    _lib.stream()), "sample_positions")
    return real_bins, rays_t, xyzs


def jitter(uniform: Optional[torch.Tensor], N: int, T: int, kind: int, device=None) -> torch.Tensor:
    """Sampling positions of a training stage from one uniform [0,1) tensor (sn_rm_jitter): kind 0 = the stage-0 bins of renderer.py:262-270
    (T = num_steps[0] + 1 edges, clamped to [0,1]), kind 1 = sample_pdf's u of renderer.py:97-102.  uniform: [N,T] (contiguous) or None."""
    if uniform is not None:
        uniform = uniform.reshape(N, T)
        device = uniform.device
    out = torch.empty(N, T, device=device, dtype=torch.float32)
    _lib.check(_lib.lib().sn_rm_jitter(_lib.dev(uniform, "uniform"), N, T, int(kind), _lib.dev(out, "out"), _lib.stream()), "jitter")
    return out


class _ray_composite(Function):
    """weights [N,T], rays_t [N,T], raw [N,T,16] (grid_mlp's output), rays_d [N,3] -> (weights_sum [N], depth [N], f_image [N,31]) in one
    kernel (sn_rm_ray_composite; renderer.py:327-347 with colour = cat([geo_feat, SH(d)]), network.py:164-170); differentiable w.r.t.
    weights and raw."""

    @staticmethod
    def forward(ctx, weights, rays_t, raw, rays_d):
        w = weights.detach().contiguous().float()
        tm = rays_t.detach().contiguous().float()
        r = raw.detach().contiguous().float()
        d = rays_d.detach().contiguous().float()
        N, T = w.shape
        assert r.shape == (N, T, 16) and d.shape == (N, 3)
        ws = torch.empty(N, device=w.device, dtype=torch.float32)
        depth = torch.empty(N, device=w.device, dtype=torch.float32)
        f = torch.empty(N, 31, device=w.device, dtype=torch.float32)
        _lib.check(_lib.lib().sn_rm_ray_composite(_lib.dev(w, "weights"), _lib.dev(tm, "rays_t"), _lib.dev(r, "raw"), _lib.dev(d, "rays_d"), N, T,
                                                  _lib.dev(ws, "weights_sum"), _lib.dev(depth, "depth"), _lib.dev(f, "f_image"), _lib.stream()),
                   "ray_composite")
        ctx.save_for_backward(w, tm, r, d)
        ctx.set_materialize_grads(False)
        return ws, depth, f

    @staticmethod
    def backward(ctx, g_ws, g_depth, g_f):
        w, tm, r, d = ctx.saved_tensors
        if g_ws is None and g_depth is None and g_f is None:
            return None, None, None, None
        N, T = w.shape
        c = lambda t: t.contiguous().float() if t is not None else None        # noqa: E731
        g_ws, g_depth, g_f = c(g_ws), c(g_depth), c(g_f)
        gw = torch.empty_like(w)
        gr = torch.empty_like(r)
        _lib.check(_lib.lib().sn_rm_ray_composite_backward(_lib.dev(w, "weights"), _lib.dev(tm, "rays_t"), _lib.dev(r, "raw"), _lib.dev(d, "rays_d"),
                                                           _lib.dev(g_ws, "grad_weights_sum"), _lib.dev(g_depth, "grad_depth"), _lib.dev(g_f, "grad_f_image"),
                                                           N, T, _lib.dev(gw, "grad_weights"), _lib.dev(gr, "grad_raw"), _lib.stream()),
                   "ray_composite_backward")
        return gw, None, gr, None


def ray_composite(weights, rays_t, raw, rays_d):
    return _ray_composite.apply(weights, rays_t, raw, rays_d)


def _proposal_loss_launch(bins, w, ref_bins, ref_w, scale, scale_dev, per_ray_ptr, grad_ptr, what):
    """One stage, one direction (sn_rm_proposal_loss_long): up to 512 samples per ray the arrays of a ray sit in LDS; longer rays get a
    workspace (at most 64 MiB) -- the same kernel arithmetic either way, nothing falls back to torch."""
    N, T, Tr = w.shape[0], w.shape[1], ref_w.shape[1]
    lib = _lib.lib()
    nbytes = 0 if max(T, Tr) <= 512 else int(lib.sn_rm_proposal_loss_workspace_bytes(N, T, Tr, 0 if grad_ptr is None else 1))
    ws = torch.empty((nbytes + 7) // 8, device=w.device, dtype=torch.float64) if nbytes else None
    _lib.check(lib.sn_rm_proposal_loss_long(_lib.dev(bins, "bins"), _lib.dev(w, "weights"), _lib.dev(ref_bins, "ref_bins"), _lib.dev(ref_w, "ref_weights"),
                                            N, T, Tr, float(scale), scale_dev, per_ray_ptr, grad_ptr, ws.data_ptr() if ws is not None else None,
                                            nbytes, _lib.stream()), what)
    if ws is not None:
        ws.record_stream(torch.cuda.current_stream(w.device))


class _proposal_loss_all(Function):
    """The whole inter-level proposal loss (nerf/renderer.py:30-57) as one autograd node: one kernel per proposal stage forward and backward,
    the mean's 1 / (N Tr) and the incoming gradient scalar applied inside the kernels (sn_rm_proposal_loss_scaled)."""

    @staticmethod
    def forward(ctx, ref_bins, ref_weights, *bw):
        ref_bins, ref_w = ref_bins.detach().contiguous().float(), ref_weights.detach().contiguous().float()
        N, Tr = ref_w.shape
        S = len(bw) // 2
        bins = [b.detach().contiguous().float() for b in bw[:S]]
        ws = [w.detach().contiguous().float() for w in bw[S:]]
        per_ray = torch.empty(S, N, device=ref_w.device, dtype=torch.float32)
        scale = 1.0 / float(N * Tr)
        for k in range(S):
            _proposal_loss_launch(bins[k], ws[k], ref_bins, ref_w, scale, None, per_ray[k].data_ptr(), None, "proposal_loss")
        ctx.save_for_backward(ref_bins, ref_w, *bins, *ws)
        ctx.S = S
        return per_ray.sum()

    @staticmethod
    def backward(ctx, grad_out):
        saved = ctx.saved_tensors
        ref_bins, ref_w = saved[0], saved[1]
        S = ctx.S
        bins, ws = saved[2:2 + S], saved[2 + S:]
        N, Tr = ref_w.shape
        scale = 1.0 / float(N * Tr)
        g = grad_out.detach().reshape(1).contiguous().float()         # stays on the device: no host sync
        out = []
        for k in range(S):
            gw = torch.empty_like(ws[k])
            _proposal_loss_launch(bins[k], ws[k], ref_bins, ref_w, scale, _lib.dev(g, "grad_out"), None, _lib.dev(gw, "grad_weights"), "proposal_loss_backward")
            out.append(gw)
        return (None, None) + (None,) * S + tuple(out)


def proposal_loss_all(all_bins, all_weights):
    """sum over the proposal stages of proposal_loss_stage(...) against the last entry (renderer.py:30-57), one autograd node."""
    return _proposal_loss_all.apply(all_bins[-1], all_weights[-1], *all_bins[:-1], *all_weights[:-1])


def zeros_f32(shape, device) -> torch.Tensor:
    """torch.zeros(shape) whose fill is a library kernel on the current stream (sn_zero; capturable)."""
    t = torch.empty(shape, device=device, dtype=torch.float32)
    nbytes = t.numel() * 4
    if nbytes % 16 == 0 and t.data_ptr() % 16 == 0 and nbytes > 0:
        _lib.check(_lib.lib().sn_zero(t.data_ptr(), nbytes, _lib.stream()), "zero")
    else:
        t.zero_()
    return t


class _proposal_loss_stage(Function):
    """One proposal stage's term of the inter-level loss (nerf/renderer.py:30-57): mean over rays and final-stage
    intervals of max(w_ref - bound, 0)^2 / (w_ref + 1e-8); differentiable w.r.t. the proposal weights only (bins come
    from sample_pdf, the final stage's bins and weights are detached in the reference)."""

    @staticmethod
    def forward(ctx, bins, weights, ref_bins, ref_weights):
        bins, ref_bins = bins.detach().contiguous().float(), ref_bins.detach().contiguous().float()
        w, ref_w = weights.detach().contiguous().float(), ref_weights.detach().contiguous().float()
        N, T = w.shape
        Tr = ref_w.shape[1]
        per_ray = torch.empty(N, device=w.device, dtype=torch.float32)
        _proposal_loss_launch(bins, w, ref_bins, ref_w, 1.0, None, _lib.dev(per_ray, "loss_per_ray"), None, "proposal_loss")
        ctx.save_for_backward(bins, w, ref_bins, ref_w)
        return per_ray.sum() / float(N * Tr)

    @staticmethod
    def backward(ctx, grad_out):
        bins, w, ref_bins, ref_w = ctx.saved_tensors
        N, T = w.shape
        Tr = ref_w.shape[1]
        gw = torch.empty_like(w)
        _proposal_loss_launch(bins, w, ref_bins, ref_w, 1.0, None, None, _lib.dev(gw, "grad_weights"), "proposal_loss_backward")
        return None, gw * (grad_out / float(N * Tr)), None, None      # no host sync: the scale stays a device scalar


PROPOSAL_LOSS_MAX_T = 1 << 24       # the kernels take any ray (up to 512 samples in LDS, beyond that in a workspace); nerf/renderer.py keeps the torch statement
                                    # for CPU tensors and the tests' comparison (they lower this constant)


def proposal_loss_stage(bins, weights, ref_bins, ref_weights):
    """bins [N,T+1], weights [N,T] of a proposal stage; ref_* of the final stage -> scalar (renderer.py:37-56, one stage)."""
    return _proposal_loss_stage.apply(bins, weights, ref_bins, ref_weights)


class _distort_loss(Function):
    """Mip-NeRF-360 distortion loss of nerf/renderer.py:17-27 (the reference delegates to the third-party
    torch_efficient_distloss.eff_distloss): mean over rays of (1/3) sum w_i^2 d_i + sum_ij w_i w_j |m_i - m_j|;
    value and d/dw from one kernel, the gradient is kept for backward (bins carry no gradient)."""

    @staticmethod
    def forward(ctx, bins, weights):
        bins = bins.detach().contiguous().float()
        w = weights.detach().contiguous().float()
        N, T = w.shape
        per_ray = torch.empty(N, device=w.device, dtype=torch.float32)
        gw = torch.empty_like(w)
        _lib.check(_lib.lib().sn_rm_distort_loss(_lib.dev(bins, "bins"), _lib.dev(w, "weights"), N, T, _lib.dev(per_ray, "loss_per_ray"),
                                                 _lib.dev(gw, "grad_weights"), _lib.stream()), "distort_loss")
        ctx.save_for_backward(gw)
        ctx.n = N
        return per_ray.sum() / float(N)

    @staticmethod
    def backward(ctx, grad_out):
        (gw,) = ctx.saved_tensors
        return None, gw * (grad_out / float(ctx.n))


DISTORT_LOSS_MAX_T = 1 << 24        # any ray (up to 2048 samples in LDS, beyond that read in place); as above


def distort_loss(bins, weights):
    """bins [N,T+1], weights [N,T] -> scalar (renderer.py:17-27)."""
    return _distort_loss.apply(bins, weights)


class _mask_nll(Function):
    """-log(clamp(softmax(logits)[label], eps, 1 - eps)) per ray, value and gradient in one kernel (sn_rm_mask_nll)."""

    @staticmethod
    def forward(ctx, logits, labels, eps):
        shape = logits.shape[:-1]
        lg = logits.reshape(-1, logits.shape[-1]).contiguous().float()
        lb = labels.reshape(-1).contiguous().long()
        N, K = lg.shape
        loss = torch.empty(N, device=lg.device, dtype=torch.float32)
        grad = torch.empty_like(lg) if ctx.needs_input_grad[0] else None
        _lib.check(_lib.lib().sn_rm_mask_nll(_lib.dev(lg, "logits"), _lib.dev(lb, "labels", torch.int64), N, K, float(eps), _lib.dev(loss, "loss"),
                                             _lib.dev(grad, "grad_logits"), _lib.stream()), "mask_nll")
        ctx.save_for_backward(grad)
        ctx.lshape = logits.shape
        return loss.view(*shape, 1)

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        return (grad * grad_out.reshape(-1, 1)).view(ctx.lshape), None, None


def mask_nll(logits, labels, eps: float = 1e-6):
    """The mask-field training loss of nerf/trainer.py:419-428, per ray: logits [..., n_inst], labels [...] (int) ->
    [..., 1] = -log(gather(clamp(softmax(logits, -1), eps, 1 - eps), -1, labels[..., None])); the trainer takes its .mean()."""
    return _mask_nll.apply(logits, labels, eps)


class _composite(Function):
    """out[n,k] = sum_t w[n,t] * v[n,t,k]."""

    @staticmethod
    def forward(ctx, weights, values):
        weights = weights.contiguous().float()
        values = values.contiguous().float()
        N, T = weights.shape
        K = values.shape[-1]
        out = torch.empty(N, K, device=values.device, dtype=torch.float32)
        _lib.check(_lib.lib().sn_rm_composite(_lib.dev(weights, "weights"), _lib.dev(values, "values"), N, T, K,
                                              _lib.dev(out, "out"), _lib.stream()), "composite")
        ctx.save_for_backward(weights, values)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        weights, values = ctx.saved_tensors
        N, T = weights.shape
        K = values.shape[-1]
        grad_out = grad_out.contiguous().float()
        gv = gw = None
        if ctx.needs_input_grad[1]:
            gv = torch.empty_like(values)
            _lib.check(_lib.lib().sn_rm_composite_backward(_lib.dev(weights, "weights"), _lib.dev(grad_out, "grad_out"), N, T, K,
                                                           _lib.dev(gv, "grad_values"), _lib.stream()), "composite_backward")
        if ctx.needs_input_grad[0]:
            gw = (values * grad_out.unsqueeze(1)).sum(-1)
        return gw, gv


def composite(weights, values):
    """weights [N,T], values [N,T,K] or [N,T] -> [N,K] or [N]."""
    if values.dim() == 2:
        return _composite.apply(weights, values.unsqueeze(-1)).squeeze(-1)
    return _composite.apply(weights, values)


# --------------------------------------------------------------------------------------------
# fused renderer
# --------------------------------------------------------------------------------------------
def _fill_grid(desc: _lib.GridDesc, enc, table: torch.Tensor) -> None:
    from ..gridencoder.grid import _host_offsets
    offs = _host_offsets(enc.offsets)
    desc.embeddings = table.data_ptr()
    desc.table_dtype = _lib.SN_F16 if table.dtype == torch.float16 else _lib.SN_F32
    for i, o in enumerate(offs):
        desc.offsets[i] = o
    desc.D, desc.C, desc.L = enc.input_dim, enc.level_dim, enc.num_levels
    desc.S = float(np.float32(np.log2(enc.per_level_scale)))
    desc.H = int(enc.base_resolution)
    desc.gridtype, desc.align_corners, desc.interp = enc.gridtype_id, int(enc.align_corners), enc.interp_id


def grid_composite(weights: torch.Tensor, xyzs: torch.Tensor, encoder, bound: float = 1.0, tile_w: int = 0,
                   table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """composite(weights, encoder(xyzs, bound)) in one kernel (renderer.py:301-302 + 361): [N,T], [N,T,3] -> [N, L*C].
    Inference only: no autograd graph is recorded (training uses encoder(...) + composite(...))."""
    N, T = weights.shape
    assert xyzs.shape == (N, T, 3), f"xyzs {tuple(xyzs.shape)} does not match weights {tuple(weights.shape)}"
    L = _lib.lib()
    emb = (encoder.embeddings if table is None else table).detach()
    w = weights.detach().contiguous().float()
    x = xyzs.detach().contiguous().float()
    out = torch.empty(N, encoder.output_dim, device=w.device, dtype=torch.float32)
    desc = _lib.GridDesc()
    _fill_grid(desc, encoder, emb.contiguous())
    _lib.check(L.sn_rm_grid_composite(_lib.dev(x, "xyzs"), _lib.dev(w, "weights"), N, T, float(bound), C.byref(desc),
                                      int(tile_w), _lib.dev(out, "out"), _lib.stream()), "sn_rm_grid_composite")
    return out


FP16_SPLIT_LIMIT = 65504.0 * 0.5      # split-fp16 operands must stay below the fp16 range; half of it as the guard band


def mlp_wide_overflow() -> bool:
    """True if a matrix-core head MLP call since the last query produced a non-finite output row, i.e. an activation
    left the fp16 range of its split-fp16 arithmetic.  Reads and clears the device flag; synchronises."""
    flag = C.c_int32(0)
    _lib.check(_lib.lib().sn_mlp_wide_overflow(C.byref(flag)), "sn_mlp_wide_overflow")
    return bool(flag.value)


def mlp_forward(x: torch.Tensor, mlp, layer_norm: Optional[torch.nn.LayerNorm] = None, check_range: Optional[bool] = None) -> torch.Tensor:
    """SkipConnMLP / MLP forward [+ LayerNorm] in one matrix-core kernel (network.py:9-66, 115; the feature heads
    of renderer.py:359-385).  x [N, dim_in] -> [N, dim_out].  Inference only (no autograd graph); hidden width 256.
    check_range (default: environment SN_CHECK_RANGE=1): query the overflow flag after the call (one synchronisation) and
    raise if an activation left the fp16 range -- the inputs are run-time data, so no static bound exists for this kernel."""
    L = _lib.lib()
    x = x.detach().contiguous().float()
    layers = list(mlp.net)
    desc = _lib.MlpDesc()
    keep: list = []
    _fill_mlp(desc, layers, x.shape[-1], keep)
    leaky = getattr(mlp, "skip_layers", None) is not None        # SkipConnMLP uses LeakyReLU(0.01), MLP uses ReLU
    desc.activation = 1 if leaky else 0
    desc.skip_mask = sum(1 << int(i) for i in (getattr(mlp, "skip_layers", None) or []))
    need = int(L.sn_mlp_wide_workspace_bytes(C.byref(desc)))
    if need == 0:
        raise RuntimeError("mlp_forward: " + L.sn_last_error().decode())
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    out = torch.empty(x.shape[0], desc.dims[desc.num_layers], device=x.device, dtype=torch.float32)
    lw = lb = None
    eps = 0.0
    if layer_norm is not None:
        lw = layer_norm.weight.detach().contiguous().float()
        lb = layer_norm.bias.detach().contiguous().float()
        eps = float(layer_norm.eps)
    _lib.check(L.sn_mlp_wide_forward(C.byref(desc), _lib.dev(lw, "ln.weight"), _lib.dev(lb, "ln.bias"), eps,
                                     _lib.dev(x, "x"), x.shape[0], _lib.dev(out, "out"), ws.data_ptr(), ws.numel(),
                                     _lib.stream()), "sn_mlp_wide_forward")
    if check_range if check_range is not None else check_range_default:
        if mlp_wide_overflow():
            raise RuntimeError("mlp_forward: an activation left the fp16 range of the split-fp16 matrix-core arithmetic "
                               "(|v| >= 65504): outputs are not finite; run this head through the torch module instead")
    return out


def mask_head_fusable(encoder, mlp, T: int, E: int) -> bool:
    """Can sn_rm_mask_head take this head (renderer.py:376-385)?  fp32 3-D grid with level_dim 8, 256-wide MLP without skip
    layers and at most 32 outputs, a power-of-two number of samples per ray <= 128, at most 16 appended channels."""
    return (encoder.input_dim == 3 and encoder.level_dim == 8 and encoder.embeddings.dtype == torch.float32
            and (encoder.num_levels % 2 == 0 or E == 0) and E <= 16 and 1 <= T <= 128 and (T & (T - 1)) == 0
            and getattr(mlp, "dim_hidden", 0) == 256 and mlp.dim_out <= 32 and not (getattr(mlp, "skip_layers", None) or [])
            and mlp.dim_in == encoder.output_dim + E)


def mask_head(weights: torch.Tensor, xyzs: torch.Tensor, extra: torch.Tensor, encoder, mlp, bound: float) -> torch.Tensor:
    """composite(weights, mlp(cat([encoder(xyzs, bound), extra], -1))) in ONE kernel (renderer.py:304-305, 376-385):
    weights [N,T], xyzs [N,T,3], extra [N,T,E] -> [N, mlp.dim_out].  Inference only; see mask_head_fusable()."""
    N, T = weights.shape
    E = int(extra.shape[-1]) if extra is not None else 0
    assert xyzs.shape == (N, T, 3)
    L = _lib.lib()
    w = weights.detach().contiguous().float()
    x = xyzs.detach().contiguous().float()
    e = extra.detach().contiguous().float() if E else None
    gdesc = _lib.GridDesc()
    _fill_grid(gdesc, encoder, encoder.embeddings.detach().contiguous())
    mdesc = _lib.MlpDesc()
    keep: list = []
    _fill_mlp(mdesc, list(mlp.net), mlp.dim_in, keep)
    mdesc.activation = 1 if getattr(mlp, "skip_layers", None) is not None else 0      # SkipConnMLP: LeakyReLU(0.01)
    need = int(L.sn_rm_mask_head_workspace_bytes(C.byref(mdesc)))
    if need == 0:
        raise RuntimeError("mask_head: " + L.sn_last_error().decode())
    ws = torch.empty(need, dtype=torch.uint8, device=w.device)
    out = torch.empty(N, mdesc.dims[mdesc.num_layers], device=w.device, dtype=torch.float32)
    _lib.check(L.sn_rm_mask_head(_lib.dev(x, "xyzs"), _lib.dev(e, "extra"), _lib.dev(w, "weights"), N, T, E, float(bound),
                                 C.byref(gdesc), C.byref(mdesc), _lib.dev(out, "out"), ws.data_ptr(), ws.numel(), _lib.stream()),
               "sn_rm_mask_head")
    return out


def _fill_mlp(desc: _lib.MlpDesc, layers: Sequence[torch.nn.Linear], dim_in: int, keep: list) -> None:
    desc.num_layers = len(layers)
    desc.activation = 0
    desc.skip_mask = 0
    desc.dims[0] = dim_in
    for l, lin in enumerate(layers):
        w = lin.weight.detach().contiguous().float()
        keep.append(w)
        desc.weight[l] = w.data_ptr()
        desc.dims[l + 1] = w.shape[0]
        if lin.bias is not None:
            b = lin.bias.detach().contiguous().float()
            keep.append(b)
            desc.bias[l] = b.data_ptr()
        else:
            desc.bias[l] = None


class RenderPlan:
    """Everything sn_rm_render_rays needs from a NeRFNetwork-shaped module, built once and reused:
    the config struct (device pointers of tables / MLP weights) and a cached workspace.

    table_dtype=torch.float16 renders from half-precision copies of the hash tables (made here,
    the module keeps its fp32 parameters); arithmetic stays fp32 either way."""

    def __init__(self, model, num_steps: Sequence[int], table_dtype=torch.float32, feat_encoder=None,
                 early_stop_eps: float = 0.0, compact_live: bool = False, tuning: Optional["Tuning"] = None):
        self.keep: list = []
        self.tuning = tuning            # None: the module default `raymarching.tuning` at call time
        cfg = _lib.RenderCfg()
        S = len(num_steps)
        if not 1 <= S <= _lib.MAX_STAGES:
            raise RuntimeError(f"num_steps must have 1..{_lib.MAX_STAGES} entries")
        cfg.num_stages = S
        for k, t in enumerate(num_steps):
            cfg.num_steps[k] = int(t)

        self.copies: list = []       # (source parameter, converted copy) pairs: see refresh_tables()

        def table_of(enc):
            src = enc.embeddings.detach()
            t = src
            if src.dtype != table_dtype or not src.is_contiguous():
                t = src.to(table_dtype).contiguous()
                self.copies.append((enc.embeddings, t))
            self.keep.append(t)
            return t

        for k in range(S - 1):
            enc = model.prop_encoders[k]
            _fill_grid(cfg.prop_grid[k], enc, table_of(enc))
            _fill_mlp(cfg.prop_mlp[k], list(model.prop_mlp[k].net), model.prop_mlp[k].dim_in, self.keep)
        _fill_grid(cfg.grid, model.grid, table_of(model.grid))
        _fill_mlp(cfg.grid_mlp, list(model.grid_mlp.net), model.grid_mlp.dim_in, self.keep)
        _fill_mlp(cfg.view_mlp, list(model.view_mlp.net), model.view_mlp.dim_in, self.keep)
        cfg.sh_degree = model.view_encoder.degree
        ab = model.aabb_infer.detach().cpu().tolist()
        for i in range(6):
            cfg.aabb[i] = ab[i]
        cfg.min_near = float(model.min_near)
        cfg.bound = float(model.bound)
        cfg.contract = int(bool(model.opt.contract))
        cfg.last_sample_opaque = int(model.opt.background == "last_sample")
        cfg.bg_color = 1.0
        self.feat_dim = 0
        if feat_encoder is not None:                    # s_grid (network.py:103): f_sam accumulated inside the render
            _fill_grid(cfg.feat_grid, feat_encoder, table_of(feat_encoder))
            cfg.with_feat = 1
            self.feat_dim = feat_encoder.output_dim
        cfg.early_stop_eps = float(early_stop_eps)      # opt-in transmittance early-out of the last stage (0 = reference behaviour)
        cfg.compact_live = int(bool(compact_live))      # opt-in per-ray termination + live-sample compaction (k_final_stage_cmp)
        self.cfg = cfg
        self._range_model = model
        self._range_versions = None
        self.activation_bound = 0.0
        self.check_range()
        self.num_steps = [int(t) for t in num_steps]
        self.geo = int(getattr(model, "geom_feat_dim", list(model.grid_mlp.net)[-1].weight.shape[0] - 1))
        self.ncol = self.geo + model.view_encoder.output_dim
        self._ws: Optional[torch.Tensor] = None

    @torch.no_grad()
    def check_range(self) -> None:
        """Static range guard of the split-fp16 MLP of the final stage.  Hash-grid features are convex combinations of
        table values, so |feature| <= max|table|; a ReLU layer's outputs are bounded by the largest row L1 norm of its
        weight times the bound of its inputs.  If features, hidden activations or weights could reach the fp16 range the
        plan switches the kernel to the exact fp32 matrix-core path (cfg.mlp_exact_fp32, ~2.2x slower final stage) instead
        of risking inf / NaN.  Re-evaluated when a parameter's version counter moved (device reductions + ONE sync); called by
        render_rays right before a launch that runs the final stage -- never for skip_final calls, whose proposal stages use
        fp32 vector arithmetic only (a training step that takes its sample positions from the fused proposal stages stays free
        of host synchronisation).  sanerf_hq_amd.optim.Adam bumps the version counters of the tensors it updates."""
        m = self._range_model
        tensors = [m.grid.embeddings] + [lin.weight for lin in m.grid_mlp.net]
        versions = tuple((t.data_ptr(), t._version) for t in tensors)
        if versions == self._range_versions:
            return
        self._range_versions = versions
        layers = list(m.grid_mlp.net)
        # every reduction stays on the device; ONE host transfer (= one sync) fetches them all
        stats = [m.grid.embeddings.detach().abs().max().float()]
        stats += [lin.weight.detach().float().abs().sum(dim=1).max() for lin in layers[:-1]]   # the last layer's outputs stay fp32 accumulators
        stats += [lin.weight.detach().abs().max().float() for lin in layers]
        vals = torch.stack(stats).tolist()
        bound = worst = vals[0]
        for row_l1 in vals[1:len(layers)]:
            bound = row_l1 * bound
            worst = max(worst, bound)
        wmax = max(vals[len(layers):]) if layers else 0.0
        if any(v != v for v in vals):
            worst = float("nan")
        self.activation_bound = max(worst, wmax)
        exact = not (self.activation_bound < FP16_SPLIT_LIMIT)          # also catches NaN
        if exact and not self.cfg.mlp_exact_fp32:
            warnings.warn(f"fused render: activations of the 32-64-64-16 MLP are only bounded by {self.activation_bound:.3g} "
                          f"(>= {FP16_SPLIT_LIMIT:.0f}): using the exact fp32 matrix-core path instead of split-fp16")
        self.cfg.mlp_exact_fp32 = int(exact)

    def invalidate_range(self) -> None:
        """Forget the cached range-guard decision: the next render re-evaluates it.  For writers that change the field's parameters WITHOUT
        moving their version counters -- `param.data.copy_()` (torch_ema's copy_to / restore around an evaluation, as the reference's trainer
        uses it), raw-pointer writers, a load_state_dict into .data -- which check_range's (data_ptr, _version) key cannot see."""
        self._range_versions = None

    @torch.no_grad()
    def refresh_tables(self) -> None:
        """Re-convert the table copies (render_table_dtype != the parameters' dtype) from the live parameters, in place:
        the plan's device pointers stay valid.  A plan over fp32 tables holds no copies and this is a no-op.  (The fp16 range
        guard is evaluated by render_rays, see check_range.)"""
        for src, copy in self.copies:
            copy.copy_(src.detach())

    def workspace(self, N: int, tile_w: int, device) -> torch.Tensor:
        (self.tuning or tuning).write(self.cfg.tuning)
        need = int(_lib.lib().sn_rm_render_workspace_bytes(C.byref(self.cfg), N, tile_w))
        if self._ws is None or self._ws.numel() < need or self._ws.device != torch.device(device):
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws


def render_rays(plan: RenderPlan, rays_o, rays_d, cam_near_far=None, bg_color: float = 1.0, tile_w: int = 0,
                want: Sequence[str] = (), u_tables: Optional[Dict[int, torch.Tensor]] = None,
                bins0_table: Optional[torch.Tensor] = None, out: Optional[Dict[str, torch.Tensor]] = None,
                skip_final: bool = False, tuning: Optional["Tuning"] = None, packed: Optional[torch.Tensor] = None,
                head_input: bool = False):
    """Fused render of N rays.  Returns dict(image [N,3], depth [N], weights_sum [N]) plus the
    per-stage tensors named in `want`: 'bins', 'weights', 'sigmas', 'inds' (all stages),
    'weights_last', 'xyzs_last', 'geo_feat_last', 'f_image'; a plan built with `feat_encoder` also
    returns 'f_feat' [N, L*C] = composite(weights_last, feat_encoder(xyzs_last)).
    packed: a contiguous fp32 [N, K >= 5] buffer -- the kernels write rgb | depth | weights_sum into its first five columns (sn_render_io.out_stride)
    and the returned image / depth / weights_sum are views of it: the payload of the image all-gather without a concatenation (dist.py).
    head_input (plans with `feat_encoder`): also return 'head_input' [N, L*C + ncol + 4] = cat([f_feat, f_image, image, depth]) -- the SAM head's
    MLP input of renderer.py:366 -- written in place by the kernels (sn_render_io.head_stride); 'f_feat' / 'f_image' are then views of it."""
    rays_o, rays_d = _flat3(rays_o), _flat3(rays_d)
    N = rays_o.shape[0]
    device = rays_o.device
    io = _lib.RenderIO()
    res: Dict[str, torch.Tensor] = {} if out is None else out
    keep: List[torch.Tensor] = []

    def buf(name, shape, dtype=torch.float32):
        t = res.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device or not t.is_contiguous():
            t = torch.empty(shape, device=device, dtype=dtype)     # (a strided view left by an earlier packed= call is not reused)
            res[name] = t
        return t

    io.rays_o, io.rays_d = _lib.dev(rays_o, "rays_o"), _lib.dev(rays_d, "rays_d")
    if cam_near_far is not None:
        cnf = cam_near_far.float()
        if cnf.shape[0] == 1:
            cnf = cnf.expand(N, 2)
        cnf = cnf.contiguous()
        keep.append(cnf)
        io.cam_near_far = _lib.dev(cnf, "cam_near_far")
    io.N, io.tile_w = N, int(tile_w)
    if bins0_table is not None:      # [T0+1]: one table for all rays; [N, T0+1]: per ray (a training step's perturbed bins)
        b0 = bins0_table.to(device).contiguous().float(); keep.append(b0); io.bins0_table = b0.data_ptr()
        if b0.dim() == 2:
            if b0.shape != (N, plan.num_steps[0] + 1):
                raise RuntimeError(f"bins0_table: expected [{N}, {plan.num_steps[0] + 1}] per-ray bins, got {tuple(b0.shape)}")
            io.bins0_ray_stride = b0.shape[1]
    if u_tables:
        for k, u in u_tables.items():
            u = u.to(device).contiguous().float(); keep.append(u); io.u_table[k] = u.data_ptr()
            if u.dim() == 2:
                if u.shape != (N, plan.num_steps[k] + 1):
                    raise RuntimeError(f"u_tables[{k}]: expected [{N}, {plan.num_steps[k] + 1}] per-ray values, got {tuple(u.shape)}")
                io.u_ray_stride[k] = u.shape[1]
    S = plan.cfg.num_stages
    want = set(want)
    if skip_final:                   # proposal stages only: the last stage's resampled bins [N, T_last+1] are the result
        if S < 2:
            raise RuntimeError("render_rays(skip_final=True) needs a schedule with proposal stages")
        io.skip_final = 1
        want = set()
        io.bins[S - 1] = buf(f"bins{S - 1}", (N, plan.num_steps[S - 1] + 1)).data_ptr()
    else:
        plan.check_range()           # fp16 range guard of the final stage's MLP: a no-op unless a parameter version moved
        if packed is not None:
            if not (packed.is_cuda and packed.dtype == torch.float32 and packed.dim() == 2 and packed.shape[0] == N and packed.shape[1] >= 5
                    and packed.is_contiguous() and packed.device == device):
                raise RuntimeError(f"render_rays: packed must be a contiguous fp32 [{N}, >=5] tensor on {device}, got {tuple(packed.shape)} {packed.dtype}")
            base = packed.data_ptr()
            io.image, io.depth, io.weights_sum, io.out_stride = base, base + 12, base + 16, packed.shape[1]
            res["image"], res["depth"], res["weights_sum"] = packed[:, :3], packed[:, 3], packed[:, 4]
        else:
            io.image = _lib.dev(buf("image", (N, 3)), "image")
            io.depth = _lib.dev(buf("depth", (N,)), "depth")
            io.weights_sum = _lib.dev(buf("weights_sum", (N,)), "weights_sum")
    for k in range(S):
        T = plan.num_steps[k]
        if "bins" in want:
            io.bins[k] = buf(f"bins{k}", (N, T + 1)).data_ptr()
        if "weights" in want or (k == S - 1 and "weights_last" in want):
            io.weights[k] = buf(f"weights{k}", (N, T)).data_ptr()
        if "sigmas" in want:
            io.sigmas[k] = buf(f"sigmas{k}", (N, T)).data_ptr()
        if "inds" in want and k >= 1:
            io.inds[k] = buf(f"inds{k}", (N, T + 1), torch.int32).data_ptr()
    Tl = plan.num_steps[S - 1]
    if "xyzs_last" in want:
        io.xyzs_last = buf("xyzs_last", (N, Tl, 3)).data_ptr()
    if "geo_feat_last" in want:
        io.geo_feat_last = buf("geo_feat_last", (N, Tl, plan.geo)).data_ptr()
    if head_input and plan.cfg.with_feat and not skip_final:
        S_head = plan.feat_dim + plan.ncol + 4
        hb = buf("head_input", (N, S_head))
        io.f_feat, io.f_image, io.head_stride = hb.data_ptr(), hb.data_ptr() + 4 * plan.feat_dim, S_head
        res["f_feat"], res["f_image"] = hb[:, :plan.feat_dim], hb[:, plan.feat_dim:plan.feat_dim + plan.ncol]
    else:
        if "f_image" in want:
            io.f_image = buf("f_image", (N, plan.ncol)).data_ptr()
        if plan.cfg.with_feat and not skip_final:
            io.f_feat = buf("f_feat", (N, plan.feat_dim)).data_ptr()
    eff = tuning or plan.tuning or globals()["tuning"]           # per call > per plan > process default
    eff.write(plan.cfg.tuning)
    need = int(_lib.lib().sn_rm_render_workspace_bytes(C.byref(plan.cfg), N, int(tile_w)))
    ws = plan.workspace(N, int(tile_w), device)
    if ws.numel() < need:                                          # (a per-call tuning that needs more than the plan's own)
        plan._ws = ws = torch.empty(need, dtype=torch.uint8, device=device)
    eff.write(plan.cfg.tuning)
    io.workspace, io.workspace_bytes = ws.data_ptr(), ws.numel()
    plan.cfg.bg_color = float(bg_color)
    _lib.check(_lib.lib().sn_rm_render_rays(C.byref(plan.cfg), C.byref(io), _lib.stream()), "render_rays")
    if "weights_last" in want:
        res["weights_last"] = res[f"weights{S - 1}"]
    return res
