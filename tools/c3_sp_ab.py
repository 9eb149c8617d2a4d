#!/usr/bin/env python3
"""400x400 (BASELINE configs[2] size: 625 workgroups of 256 rays = 1.2 rounds of the chip) through the one-lane-per-ray kernels
(tile order / linear order) and through the several-lanes-per-ray kernels with their ray-count thresholds raised:
usage (GPU box): python tools/c3_sp_ab.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    import torch
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
        sys.path.insert(0, p)
    from bench_configs import build, timeit
    from sanerf_hq_amd import raymarching as rm, synth
    import _tuning
    _tuning.apply_default()
    dev = torch.device("cuda:0")
    model = build(False, False, dev).eval()
    H = W = 400
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
    plan = model._get_plan()
    tw = W if sys.argv[1] == "tile" else 0
    with torch.no_grad():
        ref = rm.render_rays(plan, ro, rd, tile_w=W)["image"].clone()
        t = timeit(lambda: rm.render_rays(plan, ro, rd, tile_w=tw), 3, 20)
        img = rm.render_rays(plan, ro, rd, tile_w=tw)["image"]
    print(f"{sys.argv[1]:6s} tuning {os.environ.get('SN_TUNING', '-'):48s} {t * 1e3:.3f} ms  bit-equal to the tile-order image: {bool(torch.equal(img, ref))}")
else:
    for mode, env in (("tile", {}), ("linear", {}), ("linear", {"SN_TUNING": "prop_sp_max_rays=200000"}), ("linear", {"SN_TUNING": "final_sp_max_rays=200000"}),
                      ("linear", {"SN_TUNING": "prop_sp_max_rays=200000,final_sp_max_rays=200000"})):
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=e)
