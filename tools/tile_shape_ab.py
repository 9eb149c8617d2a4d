#!/usr/bin/env python3
"""Wave-tile shape A/B (round-4 verdict item 3): the 64 lanes of a wave cover 2^w x 2^(6-w) pixels (sn_render_tuning.wave_tile).  The hash is
linear in x (x ^ y P1 ^ z P2, gridencoder.cu:62-79), so which image axis runs along the lanes decides how many 128-byte lines a gather
instruction touches.  Prints ms per shape for the bench schedules and checks the images are bit-equal to the default's.
usage (GPU box, repo root): python tools/tile_shape_ab.py [hw]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import timeit  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402

dev = torch.device("cuda:0")
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 800
H = W = hw
shapes = {0: "8x8 (default)", 1: "2x32", 2: "4x16", 4: "16x4", 5: "32x2"}
for az in (30.0, 120.0):                                       # two camera azimuths: which world axis the image x axis follows changes with the view
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, az), synth.pinhole_intrinsics(H, W), H, W, device=dev)
    for steps, name in (([128], "flat128"), ([128, 64, 32], "ref")):
        params = synth.synthetic_params(steps, seed=0)
        model = synth.product_model(params, steps, False, dev)
        for tables in (torch.float16, torch.float32):
            plan = rm.RenderPlan(model, steps, tables)
            base = None
            row = []
            for wt, label in shapes.items():
                tn = rm.Tuning(wave_tile=wt)
                out = rm.render_rays(plan, ro, rd, tile_w=W, tuning=tn, out={})
                img = out["image"].clone()
                if base is None:
                    base = img
                same = bool(torch.equal(img, base))
                t = timeit(lambda: rm.render_rays(plan, ro, rd, tile_w=W, tuning=tn), 3, 12) * 1e3
                row.append(f"{label} {t:.3f} ms{'' if same else ' (IMAGE DIFFERS)'}")
            print(f"azimuth {az:5.1f}  {name:8s} {'f16' if tables == torch.float16 else 'f32'}  " + " | ".join(row), flush=True)
