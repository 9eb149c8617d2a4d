#!/usr/bin/env python3
"""Row bands on two streams for SMALL images without a feature stage (sn_render_tuning.band_streams: 1 never, 2 two bands, K > 2: K bands; 0 = automatic,
which takes bands from 2048 workgroups): whole-render time, images bit-equal.  The last stage holds 2 workgroups per CU (512 in flight): 625 workgroups
(400x400) are two rounds of which the second fills 22 % of the chip.  One JSON line.   usage: band_small_ab.py [sizes ...] (default 304 352 400 448 512 608)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import timeit  # noqa: E402
from helpers import product_model, synthetic_params  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402
dev = torch.device("cuda:0")
steps = [128, 64, 32]
model = product_model(synthetic_params(steps, seed=1), steps, False, dev)
sizes = [int(a) for a in sys.argv[1:]] or [304, 352, 400, 448, 512, 608]
out = {}
for S in sizes:
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(S, S), S, S, device=dev)
    for dt, tag in ((torch.float32, "f32"), (torch.float16, "f16")):
        plan = rm.RenderPlan(model, steps, dt)
        base = None
        for bands in (1, 2, 3, 4, 0):
            tu = rm.Tuning(band_streams=bands)
            full = lambda: rm.render_rays(plan, ro, rd, tile_w=S, tuning=tu)                                  # noqa: E731
            t_full = min(timeit(full, 3, 10) for _ in range(3)) * 1e3
            img = full()["image"].clone()
            if base is None:
                base = img
            out.setdefault(f"{S}x{S}_{tag}", {})[{1: "one_stream", 0: "auto"}.get(bands, f"{bands}_bands")] = {"render_ms": round(t_full, 4), "bit_equal": bool(torch.equal(img, base))}
print(json.dumps(out))
