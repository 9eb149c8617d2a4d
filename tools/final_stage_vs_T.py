#!/usr/bin/env python3
"""Time of a single-stage 800x800 render as a function of the samples per ray (fp16 tables): separates the per-sample cost of the final
stage from what a call pays once (pack kernels, launch, per-workgroup prologue / colour head): usage (GPU box): python tools/final_stage_vs_T.py"""
import os
import sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import timeit
from helpers import product_model, synthetic_params
from sanerf_hq_amd import raymarching as rm, synth
dev = torch.device("cuda:0")
H = W = 800
ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
for steps in ([16], [32], [64], [128], [256]):
    model = product_model(synthetic_params(steps, seed=0), steps, False, dev)
    plan = rm.RenderPlan(model, steps, torch.float16)
    t = timeit(lambda: rm.render_rays(plan, ro, rd, tile_w=W), 3, 10)
    print(f"steps={steps} {t*1e3:.3f} ms  per sample-step {t*1e6/steps[0]:.2f} us")
