#!/usr/bin/env python3
"""sn_render_tuning.mlp_mode A/B of the last stage's 32-64-64-16 MLP on the bench route (fp16 tables): split-fp16 x3 (default, fp32-class),
x2 (weights exact to 2^-22, activations rounded to fp16) and x1 (plain fp16 operands, fp32 accumulation).  For each mode: frame time of the
800x800 [128] and [128,64,32] renders, and the distance of its image from the reference's own output on the fp16-table fixtures
(tests/golden/render_flat128_h.npz, render_sref_h.npz; stress-init weights) -- the north star's bar is 1e-4 on RGB.  One JSON line."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import timeit  # noqa: E402
from helpers import golden, params_from_spec, product_model, spec_of, synthetic_params  # noqa: E402
from sanerf_hq_amd import _lib, raymarching as rm, synth  # noqa: E402

dev = torch.device("cuda:0")
MODES = (("f16x3", _lib.MLP_AUTO), ("f16x1", _lib.MLP_F16X1))      # (the two-product form of round 6 is in profiles/r06/mlp_modes_ab.json)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
out = {}
# ---- parity on the reference fixtures (the reference's own Python on tables of fp16 values) ----
for name, steps in (("render_flat128_h", [128]), ("render_sref_h", [128, 64, 32])):
    g = golden(name)
    model = product_model(params_from_spec(spec_of(g), tables_f16=True), steps, False, dev)
    u_tables = {k: T(g[f"u{k}"]) for k in range(1, len(steps))} if len(steps) > 1 else None
    plan = rm.RenderPlan(model, steps, torch.float16)
    for tag, mode in MODES:
        for densify in (2, 1):
            o = rm.render_rays(plan, T(g["rays_o"]), T(g["rays_d"]), u_tables=u_tables, out={},
                               tuning=rm.Tuning(mlp_mode=mode, densify=densify, final_sp_max_rays=-1, prop_sp_max_rays=-1))
            out.setdefault(name, {})[f"{tag}{'_densified' if densify == 2 else ''}"] = {
                "kernel": rm.last_launch_info()["final_kernel"],
                "max_abs_rgb_vs_reference": float(np.abs(o["image"].cpu().numpy() - g["image"]).max()),
                "max_abs_depth_vs_reference": float(np.abs(o["depth"].cpu().numpy() - g["depth"]).max()),
                "max_abs_wsum_vs_reference": float(np.abs(o["weights_sum"].cpu().numpy() - g["weights_sum"]).max())}
# ---- frame times and image distance on the bench scene ----
H = W = 800
ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
for sname, steps in (("flat128", [128]), ("ref", [128, 64, 32])):
    model = product_model(synthetic_params(steps, seed=0), steps, False, dev)
    plan = rm.RenderPlan(model, steps, torch.float16)
    base = None
    for tag, mode in MODES:
        tu = rm.Tuning(mlp_mode=mode)
        fn = lambda: rm.render_rays(plan, ro, rd, tile_w=W, tuning=tu)      # noqa: E731
        ms = min(timeit(fn, 3, 20) for _ in range(3)) * 1e3
        img = fn()["image"].clone()
        if base is None:
            base = img
        out.setdefault("bench_scene_800x800_" + sname, {})[tag] = {"ms": round(ms, 3), "rays_per_s": round(H * W / ms * 1e3, 1),
                                                                   "kernel": rm.last_launch_info()["final_kernel"],
                                                                   "max_abs_rgb_vs_f16x3": float((img - base).abs().max())}
print(json.dumps(out))
