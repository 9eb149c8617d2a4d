#!/usr/bin/env python3
"""Two samples per lane at once in the proposal stages (sn_render_tuning.prop_pair: 1 = one sample, 2 = two, 0 = automatic): frame time of an
image through the whole fused render and through the proposal stages only (skip_final), images / resampled bins bit-equal.  One JSON line.
usage: prop_pair_ab.py [sizes ...]   (default 128 200 256 304 352 400 800)"""
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
sizes = [int(a) for a in sys.argv[1:]] or [128, 200, 256, 304, 352, 400, 800]
out = {}
for S in sizes:
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(S, S), S, S, device=dev)
    for dt, tag in ((torch.float32, "f32"), (torch.float16, "f16")):
        plan = rm.RenderPlan(model, steps, dt)
        base = None
        for pair in (1, 2, 0):
            tu = rm.Tuning(prop_pair=pair)
            full = lambda: rm.render_rays(plan, ro, rd, tile_w=S, tuning=tu)                                  # noqa: E731
            prop = lambda: rm.render_rays(plan, ro, rd, tile_w=S, tuning=tu, skip_final=True, out={})         # noqa: E731
            t_full = min(timeit(full, 3, 10) for _ in range(3)) * 1e3
            t_prop = min(timeit(prop, 3, 10) for _ in range(3)) * 1e3
            img = full()["image"].clone(); b2 = prop()["bins2"].clone()
            if base is None:
                base = (img, b2)
            out.setdefault(f"{S}x{S}_{tag}", {})[{1: "one_sample", 2: "two_samples", 0: "auto"}[pair]] = {
                "render_ms": round(t_full, 4), "proposal_stages_ms": round(t_prop, 4), "bit_equal": bool(torch.equal(img, base[0]) and torch.equal(b2, base[1]))}
print(json.dumps(out))
