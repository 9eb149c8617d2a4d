"""Volume renderer with the reference's `NeRFRenderer` contract (nerf/renderer.py:142-385):
`render(rays_o, rays_d, staged=False, cam_near_far=None, **kw)` / `run(...)` returning the
same result keys (`image`, `depth`, `weights_sum`, `samvit`, `instance_mask_logits`,
`weights`, `num_points`, `proposal_loss`, `distort_loss`), same buffers (`aabb_train`,
`aabb_infer`) and the same `opt` fields.

Two execution paths:
  * fused (default whenever no gradient has to reach the radiance field): one
    `raymarching.render_rays` call = sn_rm_render_rays, which runs proposal resampling, the
    hash-grid field, the MLPs and compositing on the GPU without materialising per-sample
    tensors; the SAM-feature and mask heads then consume its last-stage samples.
  * differentiable (RGB training, perturb=True): the stage loop in torch autograd with the
    HIP encoders / `sample_pdf` / `composite` as building blocks.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import raymarching as rm


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.05):
    return rm.near_far_from_aabb(rays_o, rays_d, aabb, min_near)


def contract(x):
    return rm.contract(x)


def sample_pdf(bins, weights, T, perturb=False):
    return rm.sample_pdf(bins, weights, T, perturb)


def proposal_loss(all_bins, all_weights):
    """Inter-level proposal loss (nerf/renderer.py:30-57)."""
    if (len(all_bins) > 1 and all(w.is_cuda and w.dim() == 2 and w.shape[-1] <= rm.PROPOSAL_LOSS_MAX_T for w in all_weights)):
        return rm.proposal_loss_all(all_bins, all_weights)       # one autograd node, one kernel per stage and direction
    ref_bins = all_bins[-1].detach()
    ref_w = all_weights[-1].detach()
    total = 0
    for bins, w in zip(all_bins[:-1], all_weights[:-1]):
        if w.is_cuda and max(w.shape[-1], ref_w.shape[-1]) <= rm.PROPOSAL_LOSS_MAX_T and w.dim() == 2:
            total = total + rm.proposal_loss_stage(bins, w, ref_bins, ref_w)      # one kernel forward, one backward
            continue
        cum = torch.cat([torch.zeros_like(w[..., :1]), torch.cumsum(w, dim=-1)], dim=-1)
        last = w.shape[-1] - 1
        lo = (torch.searchsorted(bins[..., :-1].contiguous(), ref_bins[..., :-1].contiguous(), right=True) - 1).clamp(0, last)
        hi = torch.searchsorted(bins[..., 1:].contiguous(), ref_bins[..., 1:].contiguous(), right=True).clamp(0, last)
        bound = torch.take_along_dim(cum[..., 1:], hi, dim=-1) - torch.take_along_dim(cum[..., :-1], lo, dim=-1)
        total = total + ((ref_w - bound).clamp(min=0) ** 2 / (ref_w + 1e-8)).mean()
    return total


def distort_loss(bins, weights):
    """Mip-NeRF-360 distortion loss, O(T) per ray (what the reference gets from the third-party
    `eff_distloss`, nerf/renderer.py:17-27): sum_ij w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 d_i, mean over rays."""
    if weights.is_cuda and weights.dim() == 2 and weights.shape[-1] <= rm.DISTORT_LOSS_MAX_T:
        return rm.distort_loss(bins, weights)              # value and gradient from one kernel
    d = bins[..., 1:] - bins[..., :-1]
    m = bins[..., :-1] + d / 2
    wm = weights * m
    cw = torch.cumsum(weights, dim=-1) - weights          # exclusive prefix
    cwm = torch.cumsum(wm, dim=-1) - wm
    inter = 2 * (wm * cw - weights * cwm).sum(-1)
    intra = (weights * weights * d).sum(-1) / 3
    return (inter + intra).mean()


class NeRFRenderer(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.real_bound = opt.bound                         # world-space marching bound
        self.bound = 2 if opt.contract else opt.bound       # grid-query bound (renderer.py:152-155)
        self.cascade = 1 + math.ceil(math.log2(self.bound))
        self.min_near = opt.min_near
        self.density_thresh = opt.density_thresh
        b = float(self.real_bound)
        aabb = torch.FloatTensor([-b, -b, -b, b, b, b])
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())
        self._plan = None
        self._plan_key = None
        self.render_table_dtype = torch.float32             # torch.float16: render from half-precision table copies
        self.render_tables_static = False                   # True: skip the per-call refresh of those copies (frozen field)
        self.unstaged_cam_near_far = False                  # True: honour cam_near_far when staged=False (the reference ignores it)

    def forward(self, x, d, **kwargs):
        raise NotImplementedError()

    def density(self, x, **kwargs):
        raise NotImplementedError()

    def update_aabb(self, aabb):
        if not torch.is_tensor(aabb):
            aabb = torch.as_tensor(aabb).float()
        self.aabb_train = aabb.clamp(-self.real_bound, self.real_bound).to(self.aabb_train.device)
        self.aabb_infer = self.aabb_train.clone()
        self._plan = None
        print(f"[INFO] update_aabb: {self.aabb_train.cpu().numpy().tolist()}")

    # ---------------------------------------------------------------------------------------
    def render(self, rays_o, rays_d, staged=False, cam_near_far=None, **kwargs):
        if not staged:
            # the reference drops cam_near_far here (renderer.py:187-188: `self.run(rays_o, rays_d, **kwargs)`), so every
            # un-staged call -- all of its training steps -- marches the full aabb interval; same here unless opted in
            if getattr(self, "unstaged_cam_near_far", False):
                return self.run(rays_o, rays_d, cam_near_far=cam_near_far, **kwargs)
            return self.run(rays_o, rays_d, **kwargs)
        # staged inference (renderer.py:185-219): chunks of max_ray_batch, results scattered into place
        N = rays_o.shape[0]
        results: Dict[str, torch.Tensor] = {}
        step = self.opt.max_ray_batch
        for head in range(0, N, step):
            tail = min(head + step, N)
            cnf = cam_near_far if cam_near_far is None or cam_near_far.shape[0] == 1 else cam_near_far[head:tail]
            part = self.run(rays_o[head:tail], rays_d[head:tail], cam_near_far=cnf, _image_rays=N, **kwargs)   # (one route for every chunk of the image)
            for k, v in part.items():
                if v is None:
                    continue
                if torch.is_tensor(v):
                    if k not in results:
                        results[k] = torch.empty(N, *v.shape[1:], device=rays_o.device)
                    results[k][head:tail] = v
                else:
                    results[k] = v
        return results

    # ---------------------------------------------------------------------------------------
    def _core_parameters(self):
        for name, p in self.named_parameters():
            if name.startswith(("grid", "view_mlp", "prop_")):
                yield p

    def _needs_field_grad(self, update_proposal: bool) -> bool:
        if not torch.is_grad_enabled():
            return False
        return any(p.requires_grad for p in self._core_parameters())

    def _load_from_state_dict(self, *args, **kwargs):
        # a checkpoint load rewrites tables / aabb in place (same data_ptr): never reuse a plan across it
        self._plan = None
        return super()._load_from_state_dict(*args, **kwargs)

    def invalidate_render_plan(self):
        """Drop the cached fused-render plan (call after writing parameters through `.data`, e.g. an EMA copy_to)."""
        self._plan = None

    def _plan_tensors(self, with_feat: bool):
        encs = [self.grid] + list(getattr(self, "prop_encoders", [])) + ([self.s_grid] if with_feat else [])
        mlps = [self.grid_mlp, self.view_mlp] + list(getattr(self, "prop_mlp", []))
        return [e.embeddings for e in encs] + [lin.weight for m in mlps for lin in m.net]

    def _get_plan(self, with_feat: bool = False):
        """The cached RenderPlan.  The plan holds device pointers into the module's own fp32 tensors (in-place updates
        by an optimiser step / load_state_dict are seen as they are), so it is rebuilt only when a tensor moved
        (data_ptr / device / dtype) or the schedule changed.  What the plan COPIES is refreshed on every call: the aabb
        (host values, memoised on the buffer's version counter) and, with render_table_dtype=float16, the
        half-precision table copies (one conversion pass per render, ~20 us for the main grid; writes through `.data`
        do not bump version counters, so nothing cheaper is safe -- set `render_tables_static = True` for a frozen
        field to skip it)."""
        tensors = self._plan_tensors(with_feat)
        key = (tuple((t.data_ptr(), t.dtype, t.device) for t in tensors), tuple(self.opt.num_steps), self.render_table_dtype,
               with_feat, float(getattr(self.opt, "early_stop_eps", 0.0) or 0.0), bool(getattr(self.opt, "compact_live", False)))
        if self._plan is None or self._plan_key != key:
            self._plan = rm.RenderPlan(self, self.opt.num_steps, self.render_table_dtype,
                                       feat_encoder=self.s_grid if with_feat else None,
                                       early_stop_eps=float(getattr(self.opt, "early_stop_eps", 0.0) or 0.0),
                                       compact_live=bool(getattr(self.opt, "compact_live", False)) and self._fused_kind() == "main")
            self._plan_key = key
        elif not getattr(self, "render_tables_static", False):
            self._plan.refresh_tables()        # (the fp16 range guard is re-checked by render_rays before a final-stage launch)
        ab = rm._host_values(self.aabb_train if self.training else self.aabb_infer)
        for i in range(6):
            self._plan.cfg.aabb[i] = ab[i]
        return self._plan

    # A subclass whose forward() is NOT "grid -> grid_mlp -> [trunc_exp(sigma) | geo_feat], colour = view_mlp(cat(geo_feat, SH(d)))"
    # (network.py:146-186) sets this to False: the fused kernels restate that structure, they do not call forward().
    standard_field = True
    fused_mask_head = True        # inference: sn_rm_mask_head (one kernel); False = the three-kernel route (A/B, tests)
    fused_proposals = True        # training calls whose proposal networks get no gradient run those stages in the fused kernels
    fused_min_rays = 16384        # fields of other sizes than the reference network's: batches below this take the operator chain (see run())

    def _fused_shape(self) -> bool:
        """Can sn_rm_render_rays take this field?  Two kernels serve the last stage: the one instantiated for the reference
        network's own sizes (nerf/network.py:93-98, 131-143: L=16 F=2 grid, 32-64-64-16 and 31-32-32-3 bias-free ReLU MLPs) on the
        matrix cores, and a size-agnostic one (k_final_stage_any) for any other field of the same structure -- level_dim 2,
        up to 32 levels' worth of 64 features, bias-free ReLU MLPs of <= 4 layers and <= 64 neurons, <= 31 geometry channels;
        BASELINE configs[0] (L=8 grid, 16-32-16 / 31-32-3 MLPs) is one.  Proposal stages must be the reference's (L=5 F=2 grids,
        10-16-1 MLPs).  Anything else renders through the stage loop over the stand-alone HIP operators (grid_encode, SH,
        sample_pdf, weights, composite); the library itself would answer SN_ERR_UNSUPPORTED."""
        return self._fused_kind() is not None

    def _fused_kind(self):
        def dims(mlp):
            net = list(getattr(mlp, "net", []))
            return [net[0].weight.shape[1]] + [l.weight.shape[0] for l in net] if net and all(l.bias is None for l in net) else None
        try:
            if not self.standard_field:
                return None
            g = self.grid
            if not (g.input_dim == 3 and g.level_dim == 2 and getattr(self.view_encoder, "degree", 0) == 4):
                return None
            n_prop = len(self.opt.num_steps) - 1
            if n_prop > 0:
                encs, mlps = list(self.prop_encoders), list(self.prop_mlp)
                if not (len(encs) >= n_prop and all(
                        e.input_dim == 3 and e.num_levels == 5 and e.level_dim == 2 and e.gridtype_id == 0 and not e.align_corners
                        and e.interp_id == 0 and dims(m) == [10, 16, 1] for e, m in zip(encs[:n_prop], mlps[:n_prop]))):
                    return None
            dg, dv = dims(self.grid_mlp), dims(self.view_mlp)
            if (g.num_levels == 16 and g.gridtype_id == 0 and not g.align_corners and g.interp_id == 0
                    and dg == [32, 64, 64, 16] and dv == [31, 32, 32, 3]):
                return "main"
            from .network import MLP                                                            # network.py:9-29: ReLU between the layers
            relu = all(type(m) is MLP for m in (self.grid_mlp, self.view_mlp))                  # (a subclass may change forward(): exact type)
            if (dg is not None and dv is not None and relu and g.num_levels * 2 <= 64 and len(dg) <= 5 and len(dv) <= 5
                    and max(dg) <= 64 and max(dv) <= 64 and dg[0] == g.num_levels * 2 and 2 <= dg[-1] <= 32
                    and dv[0] == dg[-1] - 1 + 16 and dv[-1] == 3 and getattr(self, "geom_feat_dim", dg[-1] - 1) == dg[-1] - 1):
                return "any"
            return None
        except AttributeError:
            return None

    def _sam_fusable(self) -> bool:
        """f_sam can be accumulated inside the fused render (no graph through s_grid wanted, standard hash grid)."""
        if not self.opt.with_sam or not self.opt.sam_use_view_direction:
            return False
        if torch.is_grad_enabled() and self.s_grid.embeddings.requires_grad:
            return False
        if self._fused_kind() != "main":                    # the in-render feature stage is instantiated next to the reference network's last stage only
            return False
        g = self.s_grid
        return g.gridtype_id == 0 and not g.align_corners and g.interp_id == 0 and g.level_dim in (2, 4, 8)

    def run(self, rays_o, rays_d, bg_color=None, perturb=False, cam_near_far=None, update_proposal=True,
            return_feats=0, return_mask=0, H=None, W=None, tile_w=0, **kwargs):
        if self.opt.render_mesh:
            return {}                                       # the reference's mesh branch is commented out (renderer.py:257,386)
        if bg_color is None:
            bg_color = 1
        kind = self._fused_kind()
        # the size-agnostic last stage gives a ray to one lane for all its samples: below ~16 k rays the chip is mostly idle and the operator
        # chain, which spreads the SAMPLES over the lanes, is faster (configs[0], 4096 rays: 0.63 vs 0.44 ms; 160 000 rays: 1.9 vs 3.2 ms)
        # (single-stage fields only: with proposal stages in front, the chain would also leave THEIR fused kernels)
        if kind == "any" and len(self.opt.num_steps) == 1 and int(kwargs.get("_image_rays", rays_o.shape[0])) < self.fused_min_rays:
            kind = None
        if perturb or self._needs_field_grad(update_proposal) or kind is None:
            return self._run_autograd(rays_o, rays_d, bg_color, perturb, cam_near_far, update_proposal,
                                      return_feats, return_mask, H, W)
        return self._run_fused(rays_o, rays_d, bg_color, cam_near_far, return_feats, return_mask, H, W, tile_w, packed=kwargs.get("packed"))

    # ---------------------------------------------------------------------------------------
    @staticmethod
    def _head_mlp(seq, x):
        """nn.Sequential(SkipConnMLP[, LayerNorm]) of a feature head.  Without autograd (inference) the whole stack is
        one matrix-core kernel (rm.mlp_forward); with autograd it is the torch module."""
        mlp = seq[0]
        ln = seq[1] if len(seq) > 1 else None
        fusable = (not torch.is_grad_enabled() and x.is_cuda and mlp.dim_hidden == 256 and mlp.dim_out <= 256
                   and len(mlp.net) <= 8 and (ln is None or isinstance(ln, torch.nn.LayerNorm)))
        if not fusable:
            return seq(x)
        lead = x.shape[:-1]
        return rm.mlp_forward(x.reshape(-1, x.shape[-1]), mlp, ln).reshape(*lead, -1)

    def _heads(self, results, weights, xyzs, geo_feat, f_image, image, depth, return_feats, return_mask, H, W,
               tile_w=0, f_sam=None, head_input=None):
        opt = self.opt
        if opt.with_sam:                                    # renderer.py:301-302, 359-374
            if head_input is not None:
                pass                                        # cat([f_sam, f_image, image, depth]) was written in place by the fused render
            elif f_sam is not None:
                pass                                        # accumulated inside the fused render (feature stage)
            elif torch.is_grad_enabled() and any(t.requires_grad for t in (self.s_grid.embeddings, weights, xyzs)):
                features = self.s_grid(xyzs, bound=self.bound)
                f_sam = rm.composite(weights, features)
            else:   # inference: one kernel, no [N*T, 128] intermediate
                f_sam = rm.grid_composite(weights, xyzs, self.s_grid, self.bound, tile_w=tile_w)
            if head_input is not None:
                f = head_input
            elif opt.sam_use_view_direction:
                f = torch.cat([f_sam, f_image, image, depth.unsqueeze(-1)], dim=-1)
            else:
                f = torch.cat([f_sam, rm.composite(weights, geo_feat), image, depth.unsqueeze(-1)], dim=-1)
            samvit = self._head_mlp(self.samvit_mlp, f)
            if return_feats > 0:
                results["samvit"] = samvit.view(H, W, -1)
        if return_mask > 0:                                 # renderer.py:304-305, 376-385
            if opt.mask_mlp_type == "default":
                mlp = self.mask_mlp[0]
                if (not torch.is_grad_enabled() and weights.is_cuda and len(self.mask_mlp) == 1
                        and rm.mask_head_fusable(self.m_grid, mlp, weights.shape[1], geo_feat.shape[-1])
                        and self.fused_mask_head):
                    # inference: grid gather -> matrix-core MLP -> compositing in one kernel, nothing per-sample written
                    results["instance_mask_logits"] = rm.mask_head(weights, xyzs, geo_feat, self.m_grid, mlp, self.bound)
                    return
                # features and (detached) geometry channels land in one [.., 143] buffer in a single pass; in training the table gets its gradient
                # through ops._grid_encode_cat (the reference: torch.cat([m_grid(xyzs), geo_feat.detach()]), renderer.py:380)
                mlp_in = self.m_grid.forward_cat(xyzs.detach(), geo_feat, bound=self.bound)
                point_masks = self._head_mlp(self.mask_mlp, mlp_in)
            else:
                raise RuntimeError("mask_mlp_type='lightweight_mask' is dimensionally inconsistent in the reference "
                                   "(renderer.py:381 feeds 63 features into a 35-input MLP, network.py:128)")
            results["instance_mask_logits"] = rm.composite(weights.detach(), point_masks)

    def _run_fused(self, rays_o, rays_d, bg_color, cam_near_far, return_feats, return_mask, H, W, tile_w, packed=None):
        """packed: optional [N, >=5] buffer that receives rgb | depth | weights_sum in place (rm.render_rays; the all-gather payload of dist.py);
        only with a scalar background (a tensor background is added after the kernel, renderer.py:353)."""
        opt = self.opt
        need_heads = opt.with_sam or return_mask > 0
        fused_sam = self._sam_fusable()
        need_samples = return_mask > 0 or (opt.with_sam and not fused_sam)      # per-sample tensors for the torch heads
        want = []
        if opt.with_sam:
            want.append("f_image")
        if need_samples:
            want += ["weights_last", "xyzs_last"]
        if return_mask > 0 or (opt.with_sam and not opt.sam_use_view_direction):
            want.append("geo_feat_last")                    # [N,T,15] per-sample features
        plan = self._get_plan(with_feat=fused_sam)
        bg = float(bg_color) if not torch.is_tensor(bg_color) else 0.0
        # the SAM head's MLP input [N, 163] = f_sam | f_image | image | depth written in place by the kernels (no torch.cat, renderer.py:366);
        # a tensor background is blended after the kernel, so the image columns would be stale: that case keeps the concatenation
        head_in = fused_sam and opt.sam_use_view_direction and not torch.is_tensor(bg_color)
        with torch.no_grad():
            out = rm.render_rays(plan, rays_o, rays_d, cam_near_far=cam_near_far, bg_color=bg, tile_w=tile_w, want=want,
                                 packed=None if torch.is_tensor(bg_color) else packed, head_input=head_in)
            image = out["image"]
            if torch.is_tensor(bg_color):                   # per-ray / rgb background (renderer.py:353)
                image = image + (1 - out["weights_sum"]).unsqueeze(-1) * bg_color
        results = {"weights_sum": out["weights_sum"], "depth": out["depth"], "image": image}
        if need_heads:
            self._heads(results, out.get("weights_last"), out.get("xyzs_last"), out.get("geo_feat_last"), out.get("f_image"),
                        image, out["depth"], return_feats, return_mask, H, W, tile_w=tile_w or 0, f_sam=out.get("f_feat"),
                        head_input=out.get("head_input") if head_in else None)
        return results

    # ---------------------------------------------------------------------------------------
    fused_training_ops = True     # RGB-mode training: unit-cube positions, fused small MLPs, one-kernel per-ray head (False: the operator chain)

    def _unit_path_ok(self, rays_o, bg_color, return_mask) -> bool:
        """The training route built from the fused training operators (_run_autograd_unit): the reference network's own field
        (raw = [sigma_raw | 15 geometry channels], degree-4 SH, scalar background) without feature heads."""
        opt = self.opt
        return (self.fused_training_ops and rays_o.is_cuda and not torch.is_tensor(bg_color) and not opt.with_sam and not return_mask > 0
                and self._fused_kind() == "main" and hasattr(self, "field_unit") and hasattr(self, "density_unit")
                and max(opt.num_steps) <= rm.WEIGHTS_BACKWARD_MAX_T and not torch.is_autocast_enabled())

    def _run_autograd_unit(self, rays_o, rays_d, bg_color, perturb, cam_near_far, update_proposal):
        """renderer.py:261-357 for RGB-mode training out of the fused training operators: per stage one geometry kernel that emits the grid
        encoder's unit-cube coordinates, grid_encode, ONE kernel for the stage's perceptron (trunc_exp folded in), one for sigma -> weights;
        then one kernel for weights_sum / depth / f_image (SH once per ray) and one for view_mlp + sigmoid + background.  All jitter comes from
        ONE torch.rand call.  Same results as _run_autograd up to fp32 round-off (tests/test_gpu_train_ops.py)."""
        opt = self.opt
        rays_o = rays_o.contiguous()
        rays_d = rays_d.contiguous()
        N = rays_o.shape[0]
        device = rays_o.device
        steps = list(opt.num_steps)
        nears, fars = rm.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer, self.min_near)
        if cam_near_far is not None:
            nears = torch.maximum(nears, cam_near_far[:, [0]])
            fars = torch.minimum(fars, cam_near_far[:, [1]])
        rand = torch.rand(N * sum(T + 1 for T in steps), device=device) if perturb else None
        cursor = [0]

        def draw(T1):                                       # the next N*T1 uniform numbers (contiguous), or None
            if rand is None:
                return None
            r = rand[cursor[0]:cursor[0] + N * T1]
            cursor[0] += N * T1
            return r

        prop_needs_grad = update_proposal and torch.is_grad_enabled() and any(
            p.requires_grad for m in (list(self.prop_encoders) + list(self.prop_mlp)) for p in m.parameters())
        wants_prop_loss = self.training and opt.lambda_proposal > 0 and update_proposal
        all_bins, all_weights = [], []
        bins = weights = None
        first_stage = 0
        if len(steps) > 1 and not prop_needs_grad and not wants_prop_loss and self.fused_proposals:
            # proposal stages that receive no gradient: the fused inference kernels on the same jitter (see _run_autograd)
            if perturb:
                b0 = rm.jitter(draw(steps[0] + 1), N, steps[0] + 1, 0)
                u_tabs = {k: rm.jitter(draw(steps[k] + 1), N, steps[k] + 1, 1) for k in range(1, len(steps))}
            else:
                b0, u_tabs = torch.linspace(0, 1, steps[0] + 1, device=device), None
            with torch.no_grad():
                got = rm.render_rays(self._get_plan(), rays_o, rays_d, cam_near_far=cam_near_far, bins0_table=b0, u_tables=u_tabs,
                                     skip_final=True, out={})
            bins = got[f"bins{len(steps) - 1}"]
            first_stage = len(steps) - 1
        last = len(steps) - 1
        raw = rays_t = None
        for k, T in enumerate(steps):
            if k < first_stage:
                continue
            if k == first_stage and first_stage > 0:
                pass
            elif k == 0:
                bins = rm.jitter(draw(T + 1), N, T + 1, 0, device=device)
            elif perturb:
                bins = rm.sample_pdf(bins, weights, T + 1, False, u=rm.jitter(draw(T + 1), N, T + 1, 1))
            else:
                bins = rm.sample_pdf(bins, weights, T + 1, False)
            real_bins, rays_t, x01 = rm.sample_positions(rays_o, rays_d, nears, fars, bins, contract=opt.contract, grid_bound=float(self.bound))
            if k != last:
                with torch.set_grad_enabled(update_proposal and torch.is_grad_enabled()):
                    sigmas = self.density_unit(x01, k)
            else:
                sigmas, raw = self.field_unit(x01)
            weights = rm.weights_from_sigma(real_bins, sigmas, opt.background == "last_sample")
            if self.training:
                all_bins.append(bins)
                all_weights.append(weights)
        weights_sum, depth, f_image = rm.ray_composite(weights, rays_t, raw, rays_d)
        image = self.view_mlp.forward_sigmoid_bg(f_image, weights_sum, float(bg_color))
        results = {}
        if self.training and not opt.with_mask:
            results["num_points"] = N * steps[-1]
            results["weights"] = weights
            if opt.lambda_proposal > 0 and update_proposal:
                results["proposal_loss"] = proposal_loss(all_bins, all_weights)
            if opt.lambda_distort > 0:
                results["distort_loss"] = distort_loss(bins, weights)
        results["weights_sum"] = weights_sum
        results["depth"] = depth
        results["image"] = image
        return results

    def _run_autograd(self, rays_o, rays_d, bg_color, perturb, cam_near_far, update_proposal,
                      return_feats, return_mask, H, W):
        """The stage loop of renderer.py:261-357 in torch autograd over the HIP encoders."""
        if self._unit_path_ok(rays_o, bg_color, return_mask):
            return self._run_autograd_unit(rays_o, rays_d, bg_color, perturb, cam_near_far, update_proposal)
        opt = self.opt
        rays_o = rays_o.contiguous()
        rays_d = rays_d.contiguous()
        N = rays_o.shape[0]
        device = rays_o.device
        nears, fars = rm.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer, self.min_near)
        if cam_near_far is not None:
            nears = torch.maximum(nears, cam_near_far[:, [0]])
            fars = torch.minimum(fars, cam_near_far[:, [1]])

        all_bins, all_weights = [], []
        steps = opt.num_steps
        bins = weights = None
        # Proposal networks that receive no gradient in this call (trainer.py:372-373: after step 3000 they are updated on every
        # 5th step only; mask / SAM training never updates them) are pure inference: their stages run in the fused kernels --
        # per-ray perturbed bins and u as inputs, exactly the reference's jitter (renderer.py:101-102, 267-270) -- and only the
        # last stage, which carries the gradients, goes through the autograd operators below.
        prop_needs_grad = update_proposal and torch.is_grad_enabled() and any(
            p.requires_grad for m in (list(getattr(self, "prop_encoders", [])) + list(getattr(self, "prop_mlp", []))) for p in m.parameters())
        wants_prop_loss = self.training and not opt.with_mask and not opt.with_sam and opt.lambda_proposal > 0 and update_proposal
        first_stage = 0
        if (len(steps) > 1 and not prop_needs_grad and not wants_prop_loss and rays_o.is_cuda and self._fused_shape()
                and self.fused_proposals):
            b0 = torch.linspace(0, 1, steps[0] + 1, device=device)
            u_tabs = None
            if perturb:
                b0 = (b0.unsqueeze(0).expand(N, -1) + (torch.rand(N, steps[0] + 1, device=device) - 0.5) / steps[0]).clamp(0, 1)
                u_tabs = {}
                for k in range(1, len(steps)):
                    Tq = steps[k] + 1
                    base = torch.linspace(0.5 / Tq, 1 - 0.5 / Tq, steps=Tq, device=device)
                    u_tabs[k] = base.unsqueeze(0).expand(N, -1) + (torch.rand(N, Tq, device=device) - 0.5) / Tq
            with torch.no_grad():
                got = rm.render_rays(self._get_plan(), rays_o, rays_d, cam_near_far=cam_near_far, bins0_table=b0, u_tables=u_tabs,
                                     skip_final=True, out={})
            bins = got[f"bins{len(steps) - 1}"]
            first_stage = len(steps) - 1
        for k, T in enumerate(steps):
            if k < first_stage:
                continue
            if k == first_stage and first_stage > 0:
                pass                                            # bins of the last stage came from the fused proposal stages
            elif k == 0:
                # (the same stage-0 edges as the fused kernels and the fused training route: aten's scalar linspace recipe, sn_rm_jitter)
                bins = rm.jitter(torch.rand(N, T + 1, device=device) if perturb else None, N, T + 1, 0, device=device) if rays_o.is_cuda else None
                if bins is None:
                    bins = torch.linspace(0, 1, T + 1, device=device).unsqueeze(0).expand(N, -1)
                    if perturb:
                        bins = (bins + (torch.rand_like(bins) - 0.5) / T).clamp(0, 1)
            else:
                bins = rm.sample_pdf(bins, weights, T + 1, perturb)
            # bins -> distances, mid-points, (contracted) positions: nothing on this chain is differentiated
            # (sample_pdf's output carries no gradient, the encoders have no input gradient), so it is one kernel
            real_bins, rays_t, xyzs = rm.sample_positions(rays_o, rays_d, nears, fars, bins, contract=opt.contract)
            if k != len(steps) - 1:
                with torch.set_grad_enabled(update_proposal and torch.is_grad_enabled()):
                    sigmas = self.density(xyzs, proposal=k)["sigma"]
            else:
                dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
                dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)
                field = self(xyzs, dirs)
                sigmas, colors, geo_feat = field["sigma"], field["color"], field["geo_feat"]
            # renderer.py:308-325 (delta*sigma -> alpha, transmittance, weights) forward and backward in one kernel each
            weights = rm.weights_from_sigma(real_bins, sigmas, opt.background == "last_sample")
            if self.training:
                all_bins.append(bins)
                all_weights.append(weights)

        weights_sum = weights.sum(dim=-1)
        depth = (weights * rays_t).sum(dim=-1)
        f_image = rm.composite(weights, colors)
        image = torch.sigmoid(self.view_mlp(f_image))
        results = {}
        if self.training and not opt.with_mask and not opt.with_sam:
            results["num_points"] = xyzs.shape[0] * xyzs.shape[1]
            results["weights"] = weights
            if opt.lambda_proposal > 0 and update_proposal:
                results["proposal_loss"] = proposal_loss(all_bins, all_weights)
            if opt.lambda_distort > 0:
                results["distort_loss"] = distort_loss(bins, weights)
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        results["weights_sum"] = weights_sum
        results["depth"] = depth
        results["image"] = image
        if opt.with_sam or return_mask > 0:
            self._heads(results, weights, xyzs, geo_feat, f_image, image, depth, return_feats, return_mask, H, W)
        return results
