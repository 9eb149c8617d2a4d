#!/usr/bin/env python3
"""400x400 render with the mask head (renderer.py:304-305, 376-385) and the opt-in compaction scenes, for rocprofv3 kernel traces."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import build, timeit  # noqa: E402
from helpers import product_model, synthetic_params  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("SN_MASK16"):      # experiments builds: SN_MASK16=0 keeps the fused mask head on k_mlp_wide_j<3> (A/B partner of k_mask16, mlp16.inc)
    from sanerf_hq_amd import _lib
    _lib.check(_lib.lib().sn_debug_set(b"mask_head16", int(os.environ["SN_MASK16"])), "debug_set")
mode = sys.argv[1] if len(sys.argv) > 1 else "mask"
if mode == "mask":
    model = build(False, True, dev).eval()
    H = W = 400
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
    def run():
        with torch.no_grad():
            return model.render(ro, rd, staged=False, perturb=False, return_mask=1, H=H, W=W, tile_w=W)
    print("mask render ms", timeit(run, 3, 10) * 1e3)
else:   # compact: opaque field with eps, then the small-aabb scene, both through k_final_stage_cmp
    H = W = 800
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
    steps = [128]
    dense = product_model(synthetic_params(steps, seed=3, gain=40.0), steps, False, dev)
    p1 = rm.RenderPlan(dense, steps, early_stop_eps=1e-4, compact_live=True)
    print("opaque field, compact ms", timeit(lambda: rm.render_rays(p1, ro, rd, tile_w=W), 3, 10) * 1e3)
    soft = product_model(synthetic_params(steps, seed=5), steps, False, dev)
    p2 = rm.RenderPlan(soft, steps, compact_live=True)
    for i, v in enumerate([-0.25] * 3 + [0.25] * 3):
        p2.cfg.aabb[i] = v
    print("small aabb, compact ms", timeit(lambda: rm.render_rays(p2, ro, rd, tile_w=W), 3, 10) * 1e3)
