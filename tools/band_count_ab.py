#!/usr/bin/env python3
"""How many row bands on the two HIP streams (sn_render_tuning.band_streams: 1 = none, 2 = two bands, K > 2 = K bands dealt alternately to the two
streams)?  Reference schedule [128, 64, 32], fp16 and fp32 tables, 400x400 / 800x800 / 1600x1600; images checked bit-equal to the single-stream render.
usage (GPU box, repo root): python tools/band_count_ab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import timeit  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402

dev = torch.device("cuda:0")
steps = [128, 64, 32]
params = synth.synthetic_params(steps, seed=0)
model = synth.product_model(params, steps, False, dev)
for hw in (400, 800, 1600):
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(hw, hw), hw, hw, device=dev)
    for tables in (torch.float16, torch.float32):
        plan = rm.RenderPlan(model, steps, tables)
        base, row = None, []
        for k in (1, 2, 3, 4, 6, 8):
            tn = rm.Tuning(band_streams=k)
            img = rm.render_rays(plan, ro, rd, tile_w=hw, tuning=tn, out={})["image"].clone()
            if base is None:
                base = img
            same = bool(torch.equal(img, base))
            t = timeit(lambda: rm.render_rays(plan, ro, rd, tile_w=hw, tuning=tn), 3, 15) * 1e3
            row.append(f"{k}: {t:.3f}{'' if same else ' (IMAGE DIFFERS)'}")
        print(f"{hw}x{hw} {'f16' if tables == torch.float16 else 'f32'}  bands -> ms  " + " | ".join(row), flush=True)
