#!/usr/bin/env python3
"""Lanes per ray of the small-batch proposal kernels (sn_render_tuning.prop_sp_lanes: 8 / 16 / 32, 0 = automatic): frame time of a linear-order batch
through the whole fused render and through the proposal stages only (skip_final), images / resampled bins bit-equal.  One JSON line."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import timeit  # noqa: E402
from helpers import product_model, synthetic_params  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402
dev = torch.device("cuda:0")
steps = [128, 64, 32]
model = product_model(synthetic_params(steps, seed=1), steps, False, dev)
H = W = 512
roF, rdF = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 60.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
out = {}
for N in (1024, 4096, 8192, 16384):
    pix = torch.from_numpy((synth.hash_u01(N, 99) * (H * W)).astype(np.int64)).to(dev)
    ro, rd = roF[pix].contiguous(), rdF[pix].contiguous()
    for dt, tag in ((torch.float32, "f32"), (torch.float16, "f16")):
        plan = rm.RenderPlan(model, steps, dt)
        base = None
        for lanes in (8, 16, 32, 0):
            tu = rm.Tuning(prop_sp_lanes=lanes)
            full = lambda: rm.render_rays(plan, ro, rd, tuning=tu)                                  # noqa: E731
            prop = lambda: rm.render_rays(plan, ro, rd, tuning=tu, skip_final=True, out={})         # noqa: E731
            t_full = min(timeit(full, 3, 20) for _ in range(3)) * 1e3
            t_prop = min(timeit(prop, 3, 20) for _ in range(3)) * 1e3
            img = full()["image"].clone(); b2 = prop()["bins2"].clone()
            if base is None:
                base = (img, b2)
            out.setdefault(f"{N}_rays_{tag}", {})[f"lanes_{lanes or 'auto'}"] = {"render_ms": round(t_full, 4), "proposal_stages_ms": round(t_prop, 4),
                                                                               "bit_equal": bool(torch.equal(img, base[0]) and torch.equal(b2, base[1]))}
print(json.dumps(out))
