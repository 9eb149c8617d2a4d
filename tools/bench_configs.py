#!/usr/bin/env python3
"""Timings of the other BASELINE.json configurations (not the contract bench line; numbers go to DESIGN.md).
  C3: 400x400 rays + 256-d SAM-feature head (feature_container path), reference schedule [128,64,32]
  C5: mask-field training step, 4096 rays: fwd + bwd of m_grid + mask_mlp under the mask NLL, radiance field frozen
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from helpers import make_opt, synthetic_params  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402
from sanerf_hq_amd.nerf import NeRFNetwork  # noqa: E402


def timeit(fn, warm=3, iters=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def build(heads_sam, heads_mask, dev):
    params = synthetic_params([128, 64, 32], heads=True, seed=1)
    opt = make_opt(with_sam=heads_sam, with_mask=heads_mask)
    model = NeRFNetwork(opt)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    model.load_state_dict(sd, strict=False)
    return model.to(dev)


def c1_entry(dev):
    from sanerf_hq_amd.activation import trunc_exp
    from sanerf_hq_amd.encoding import get_encoder
    from sanerf_hq_amd.nerf.network import MLP
    from sanerf_hq_amd.nerf.renderer import NeRFRenderer

    class C1Field(NeRFRenderer):
        def __init__(self, opt):
            super().__init__(opt)
            self.grid, d = get_encoder("hashgrid", input_dim=3, level_dim=2, num_levels=8, log2_hashmap_size=14, desired_resolution=2048)
            self.grid_mlp = MLP(d, 16, 32, 2, bias=False)
            self.view_encoder, vd = get_encoder("sh", input_dim=3, degree=4)
            self.view_mlp = MLP(15 + vd, 3, 32, 2, bias=False)

        def forward(self, x, d, **kw):
            f = self.grid_mlp(self.grid(x, bound=self.bound))
            return dict(sigma=trunc_exp(f[..., 0]), geo_feat=f[..., 1:], color=torch.cat([f[..., 1:], self.view_encoder(d)], -1), grid_output=None)

    torch.manual_seed(5)
    model = C1Field(make_opt(num_steps=[32])).to(dev).eval()
    model.fused_min_rays = 0              # measure the fused call at both sizes (by default batches below 16 384 rays take the chain)
    with torch.no_grad():
        model.grid.embeddings.uniform_(-1.0, 1.0)
    res = {}
    for H in (64, 400):
        ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, H), H, H, device=dev)

        def render():
            with torch.no_grad():
                return model.render(ro, rd, staged=False, perturb=False, H=H, W=H, tile_w=H)
        model.standard_field = True
        t_f = timeit(render)
        img = render()["image"].clone()
        model.standard_field = False
        t_c = timeit(render)
        d = float((render()["image"] - img).abs().max())
        res[f"{H}x{H}"] = {"ms_fused_call": round(t_f * 1e3, 3), "ms_operator_chain": round(t_c * 1e3, 3), "rays_per_s_fused": round(H * H / t_f, 1),
                           "image_max_abs_diff": d}
    return res


def c3_entry(dev, warm=3, iters=10):
    """BASELINE configs[2]: 400x400 rays + 256-d SAM-feature head (feature_container path), reference schedule [128,64,32]."""
    model = build(True, False, dev).eval()
    H = W = 400
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)

    def c3():
        with torch.no_grad():
            return model.render(ro, rd, staged=False, perturb=False, return_feats=1, H=H, W=W, tile_w=W)
    # best of three timed regions: run directly after another process, the FIRST timed region of this script has shown a one-off ~25 ms stall
    # between the enqueue and the closing synchronisation (5.7 instead of 2.8 ms per frame; GPU-side event times and the CPU profile of the
    # same calls are normal, and the same function imported from another script never shows it): not the kernels' time
    t = min(timeit(c3, warm, iters) for _ in range(3))
    with torch.no_grad():
        t_rgb = timeit(lambda: rm.render_rays(model._get_plan(), ro, rd, tile_w=W), warm, iters)
    out = {"rays_per_s": round(H * W / t, 1), "ms": round(t * 1e3, 3), "rgb_only_ms": round(t_rgb * 1e3, 3)}
    # configs[2] names "same grid" as configs[1] (the fp16 configuration): tables in half (radiance, proposal and SAM-feature grids), arithmetic fp32
    model.render_table_dtype = torch.float16
    t16 = min(timeit(c3, warm, iters) for _ in range(2))
    out.update({"ms_f16_tables": round(t16 * 1e3, 3), "rays_per_s_f16_tables": round(H * W / t16, 1)})
    return out


def mask_head_entry(dev, warm=3, iters=10):
    """Mask head at inference (renderer.py:304-305, 376-385): 400x400, [128,64,32]: the one-kernel head vs the three-kernel route."""
    model = build(False, True, dev).eval()
    H = W = 400
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)

    def mask_render():
        with torch.no_grad():
            return model.render(ro, rd, staged=False, perturb=False, return_mask=1, H=H, W=W, tile_w=W)
    model.fused_mask_head = False
    t_unf = timeit(mask_render, warm, iters)
    ref_logits = mask_render()["instance_mask_logits"].clone()
    model.fused_mask_head = True
    t_fus = timeit(mask_render, warm, iters)
    dlog = float((mask_render()["instance_mask_logits"] - ref_logits).abs().max())
    with torch.no_grad():
        t_rgb = timeit(lambda: rm.render_rays(model._get_plan(), ro, rd, tile_w=W), warm, iters)
    return {"ms_fused_head": round(t_fus * 1e3, 3), "ms_three_kernel_head": round(t_unf * 1e3, 3),
            "rgb_only_ms": round(t_rgb * 1e3, 3), "max_abs_logit_diff": dlog}


def c5_setup(dev):
    model = build(False, True, dev).train()
    for n_, p in model.named_parameters():
        p.requires_grad_(n_.startswith("m_grid") or n_.startswith("mask_mlp"))
    H = W = 512
    N = 4096
    roF, rdF = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 60.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
    pix = torch.from_numpy((synth.hash_u01(N, 99) * (H * W)).astype(np.int64)).to(dev)
    ro, rd = roF[pix].contiguous(), rdF[pix].contiguous()
    labels = torch.from_numpy((synth.hash_u01(N, 100) < 0.5).astype(np.int64)).to(dev)
    return model, ro, rd, labels, N


def c5_entry(dev, warm=3, iters=10, optimisers=True):
    """BASELINE configs[4]: mask-field training step, 4096 rays (fwd + bwd of m_grid + mask_mlp under the mask NLL, field frozen)."""
    from sanerf_hq_amd.optim import Adam as HipAdam   # (csrc/optim.hip: the same dense update, one pass per tensor)
    from sanerf_hq_amd import ops
    model, ro, rd, labels, N = c5_setup(dev)
    ops.WGRAD_SIDE_STREAM = True     # opt-in: one process, no DistributedDataParallel, every MLP applied once per graph (ops.py states the contract)
    optim = HipAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=1e-15)

    def fwd_bwd():
        optim.zero_grad(set_to_none=True)
        o = model.render(ro, rd, staged=False, bg_color=1, perturb=False, update_proposal=False, return_mask=1)
        loss = rm.mask_nll(o["instance_mask_logits"], labels, 1e-6).mean()     # trainer.py:419-428 in one kernel (fwd) + one multiply (bwd)
        loss.backward()
        return loss

    def step():
        fwd_bwd()
        optim.step()
    t_fb = timeit(fwd_bwd, warm, iters)
    out = {"fwd_bwd_ms": round(t_fb * 1e3, 3), "rays_per_s_fwd_bwd": round(N / t_fb, 1),
           "fwd_bwd_single_pass_adam_ms": round(timeit(step, warm, iters) * 1e3, 3)}
    # the whole step (zero_grad, render, loss, backward, Adam) captured once as a HIP graph and replayed (sanerf_hq_amd.graph)
    from sanerf_hq_amd.graph import GraphedStep
    optim = HipAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=1e-15, capturable=True)
    g = GraphedStep(step, warmup=3)
    out["step_as_hip_graph_ms"] = round(timeit(g, warm, iters) * 1e3, 3)
    del g
    optim = HipAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=1e-15)
    if optimisers:
        optim = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=1e-15)
        out["fwd_bwd_adam_ms"] = round(timeit(step, warm, iters) * 1e3, 3)
        # opt-in touched-elements-only update (SN_ADAM_LAZY; torch.optim.SparseAdam's semantics, NOT the reference's optimiser)
        optim = HipAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=1e-15, lazy=True)
        out["fwd_bwd_lazy_adam_ms"] = round(timeit(step, warm, iters) * 1e3, 3)
        try:   # torch's single-kernel Adam over the 160 MiB table (same update rule; the reference constructs the default one)
            optim = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=1e-15, fused=True)
            out["fwd_bwd_fused_adam_ms"] = round(timeit(step, warm, iters) * 1e3, 3)
        except Exception as e:   # noqa: BLE001
            out["fwd_bwd_fused_adam_ms"] = f"unavailable: {type(e).__name__}"
    return out


def main():
    dev = torch.device("cuda:0")
    out = {}
    out["C3_sam_head_400x400"] = c3_entry(dev)
    if os.environ.get("SN_BC_ONLY") == "C3":
        print(json.dumps(out))
        return
    torch.cuda.empty_cache()
    # ---- C1 = BASELINE configs[0] (64x64, L=8 T=2^14 grid, 16-32-16 / 31-32-3 MLPs, 32 samples per ray) on the GPU: the fused call
    #      (size-agnostic last stage, k_final_stage_any) vs the stage loop over the stand-alone operators; also at 400x400 ----
    out["C1_small_field"] = c1_entry(dev)
    out["mask_head_400x400"] = mask_head_entry(dev)
    torch.cuda.empty_cache()
    # ---- opt-in early termination on an opaque field: 800x800, [128], MLP gain 40 (sigma ~0 or huge) ----
    from helpers import product_model  # noqa: E402
    steps = [128]
    dense = product_model(synthetic_params(steps, seed=3, gain=40.0), steps, False, dev)
    H = W = 800
    ro8, rd8 = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
    p_off, p_on = rm.RenderPlan(dense, steps), rm.RenderPlan(dense, steps, early_stop_eps=1e-4)
    t_off = timeit(lambda: rm.render_rays(p_off, ro8, rd8, tile_w=W))
    t_on = timeit(lambda: rm.render_rays(p_on, ro8, rd8, tile_w=W))
    d = float((rm.render_rays(p_on, ro8, rd8, tile_w=W)["weights_sum"] - rm.render_rays(p_off, ro8, rd8, tile_w=W)["weights_sum"]).abs().max())
    out["early_stop_opaque_field_800x800_flat128"] = {"ms_off": round(t_off * 1e3, 3), "ms_eps_1e-4": round(t_on * 1e3, 3), "max_abs_weights_sum_diff": d}
    # ---- live-sample compaction (k_final_stage_cmp, cfg.compact_live): per-ray termination on the same opaque field, ----
    # ---- and a scene whose aabb most rays miss (renderer.py:133-135), single stage and the reference schedule         ----
    base_ws = rm.render_rays(p_off, ro8, rd8, tile_w=W)["weights_sum"].clone()
    p_cmp, p_cmp0 = rm.RenderPlan(dense, steps, early_stop_eps=1e-4, compact_live=True), rm.RenderPlan(dense, steps, compact_live=True)
    t_cmp, t_cmp0 = timeit(lambda: rm.render_rays(p_cmp, ro8, rd8, tile_w=W)), timeit(lambda: rm.render_rays(p_cmp0, ro8, rd8, tile_w=W))
    d = float((rm.render_rays(p_cmp, ro8, rd8, tile_w=W)["weights_sum"] - base_ws).abs().max())
    out["compact_live_opaque_field_800x800_flat128"] = {"ms_default_kernel": round(t_off * 1e3, 3), "ms_wave_early_out_eps_1e-4": round(t_on * 1e3, 3),
                                                        "ms_compact_eps_1e-4": round(t_cmp * 1e3, 3), "ms_compact_nothing_to_skip": round(t_cmp0 * 1e3, 3),
                                                        "max_abs_weights_sum_diff": d}
    del dense
    box = [-0.25, -0.25, -0.25, 0.25, 0.25, 0.25]
    for sch in ([128], [128, 64, 32]):
        soft = product_model(synthetic_params(sch, seed=5), sch, False, dev)
        plans = [rm.RenderPlan(soft, sch), rm.RenderPlan(soft, sch, compact_live=True)]
        for pl in plans:
            for i in range(6):
                pl.cfg.aabb[i] = box[i]
        ts = [timeit(lambda pl=pl: rm.render_rays(pl, ro8, rd8, tile_w=W)) for pl in plans]
        o0, o1 = ({k: v.clone() for k, v in rm.render_rays(pl, ro8, rd8, tile_w=W).items()} for pl in plans)
        out["compact_live_small_aabb_800x800_" + "_".join(map(str, sch))] = {
            "ms_default_kernel": round(ts[0] * 1e3, 3), "ms_compact": round(ts[1] * 1e3, 3),
            "rays_missing_the_aabb": round(float((o1["weights_sum"] == 0).float().mean()), 4),
            "image_max_abs_diff": float((o0["image"] - o1["image"]).abs().max())}   # (per-sample third layer vs linear tail: fp32 round-off)
        del soft
    # ---- heads on an opaque field (MLP gain 40): most last-stage samples carry weight exactly 0, which the fused mask head (whole 128-sample
    #      tiles) and the in-render feature stage (wave-wide sample indices) skip -- bit-identical, see test_heads_skip_exactly_zero_weights ----
    for tag, kw, key in (("mask_head", dict(with_sam=False, with_mask=True), dict(return_mask=1)), ("sam_head", dict(with_sam=True, with_mask=False), dict(return_feats=1))):
        pm = synthetic_params([128, 64, 32], heads=True, seed=3, gain=40.0)
        mo = NeRFNetwork(make_opt(**kw))
        mo.load_state_dict({k: torch.from_numpy(v) for k, v in pm.items()}, strict=False)
        mo = mo.to(dev).eval()
        ro4, rd4 = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(400, 400), 400, 400, device=dev)

        def hr():
            with torch.no_grad():
                return mo.render(ro4, rd4, staged=False, perturb=False, H=400, W=400, tile_w=400, **key)
        with torch.no_grad():
            wl = rm.render_rays(mo._get_plan(), ro4, rd4, tile_w=400, want=("weights_last",))["weights_last"]
        out[f"opaque_field_{tag}_400x400"] = {"ms": round(timeit(hr) * 1e3, 3), "exact_zero_weights": round(float((wl == 0).float().mean()), 4)}
        del mo
    # ---- exact early-out of the march (sn_render_tuning.exact_early_out) on the opaque field: 800x800, both schedules ----
    for sch in ([128, 64, 32], [128]):
        mo = product_model(synthetic_params(sch, seed=3, gain=40.0), sch, False, dev)
        pl = rm.RenderPlan(mo, sch, torch.float16)
        ts = {nm: timeit(lambda tu=tu: rm.render_rays(pl, ro8, rd8, tile_w=W, tuning=tu)) for nm, tu in
              (("ms_default", rm.Tuning()), ("ms_last_stage_early_out_off", rm.Tuning(exact_early_out=1)), ("ms_last_stage_early_out_on", rm.Tuning(exact_early_out=2)))}
        out["exact_early_out_opaque_field_800x800_" + "_".join(map(str, sch))] = {k: round(v * 1e3, 3) for k, v in ts.items()}
        del mo, pl
    # ---- C5 ----
    out["C5_mask_training_step_4096_rays"] = c5_entry(dev)
    _, ro, rd, _, N = c5_setup(dev)
    from sanerf_hq_amd.optim import Adam as HipAdam  # noqa: E402
    # ---- RGB-mode training step (trainer.py:360-392): 4096 rays, [128,64,32], everything trainable, MSE + proposal loss ----
    torch.cuda.empty_cache()
    opt = make_opt(with_sam=False, with_mask=False)
    opt.lambda_proposal, opt.lambda_distort = 1.0, 0.0
    model = NeRFNetwork(opt)
    params = synthetic_params([128, 64, 32], seed=1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    model = model.to(dev).train()
    gt = torch.from_numpy(synth.hash_uniform((N, 3), 42, 0.0, 1.0)).to(dev)
    optim = torch.optim.Adam(model.get_params(1e-2), eps=1e-15)

    def rgb_fwd_bwd():
        optim.zero_grad(set_to_none=True)
        o = model.render(ro, rd, staged=False, bg_color=1, perturb=True, update_proposal=True)
        loss = torch.nn.functional.mse_loss(o["image"], gt) + opt.lambda_proposal * o["proposal_loss"]
        if "distort_loss" in o:                                     # main.py:112: lambda_distort defaults to 0.02
            loss = loss + opt.lambda_distort * o["distort_loss"]
        loss.backward()
        return loss

    def rgb_step():
        rgb_fwd_bwd()
        optim.step()
    t_fb = min(timeit(rgb_fwd_bwd) for _ in range(3))      # (the first measurement after construction allocates workspaces: 10+ ms)
    t_st = min(timeit(rgb_step) for _ in range(2))
    out["RGB_training_step_4096_rays"] = {"fwd_bwd_ms": round(t_fb * 1e3, 3), "fwd_bwd_adam_ms": round(t_st * 1e3, 3),
                                          "rays_per_s_step": round(N / t_st, 1)}
    # trainer.py:372-373: after step 3000 the proposal networks are updated on every 5th step only; the other four run this:
    def rgb_fwd_bwd_frozen_proposal():
        optim.zero_grad(set_to_none=True)
        o = model.render(ro, rd, staged=False, bg_color=1, perturb=True, update_proposal=False)
        torch.nn.functional.mse_loss(o["image"], gt).backward()
    out["RGB_training_step_4096_rays"]["fwd_bwd_without_proposal_update_ms"] = round(timeit(rgb_fwd_bwd_frozen_proposal) * 1e3, 3)
    # a changed loss graph / a freshly constructed optimiser allocates in its first steps (the first measurement after such a change
    # has shown 10-17 ms): best of three measurements for each of these variants
    best = lambda fn: min(timeit(fn) for _ in range(3))            # noqa: E731
    opt.lambda_distort = 0.02                                       # the reference's default loss: + distortion term
    out["RGB_training_step_4096_rays"]["fwd_bwd_with_distort_loss_ms"] = round(best(rgb_fwd_bwd) * 1e3, 3)
    opt.lambda_distort = 0.0
    optim = HipAdam(model.get_params(1e-2), eps=1e-15)             # csrc/optim.hip: one pass per tensor
    out["RGB_training_step_4096_rays"]["fwd_bwd_single_pass_adam_ms"] = round(best(rgb_step) * 1e3, 3)
    try:   # the whole step as one HIP graph (capturable Adam; perturb=True draws its jitter inside the graph)
        from sanerf_hq_amd.graph import GraphedStep
        optim = HipAdam(model.get_params(1e-2), eps=1e-15, capturable=True)
        g = GraphedStep(rgb_step, warmup=3)
        out["RGB_training_step_4096_rays"]["step_as_hip_graph_ms"] = round(best(g) * 1e3, 3)
        del g
    except Exception as e:   # noqa: BLE001
        out["RGB_training_step_4096_rays"]["step_as_hip_graph_ms"] = f"failed: {type(e).__name__}: {e}"
    optim = HipAdam(model.get_params(1e-2), eps=1e-15)
    try:   # torch's single-kernel Adam (same update rule; the reference constructs the default multi-tensor one)
        optim = torch.optim.Adam(model.get_params(1e-2), eps=1e-15, fused=True)
        out["RGB_training_step_4096_rays"]["fwd_bwd_fused_adam_ms"] = round(best(rgb_step) * 1e3, 3)
    except Exception as e:   # noqa: BLE001
        out["RGB_training_step_4096_rays"]["fwd_bwd_fused_adam_ms"] = f"unavailable: {type(e).__name__}"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
