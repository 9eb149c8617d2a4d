#!/usr/bin/env python3
"""Experiments build (SN_LIB=sanerf-hq_amd/libsanerf_hip_exp.so): the reference schedule with the last stage held to ONE workgroup per CU (84 KiB of LDS) so
that the other row band's proposal-stage workgroups share its SIMDs (tuning.experiment = EXP_FINAL_ONE_WG), against the default, for 2 and 4 bands."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import timeit  # noqa: E402
from helpers import product_model, synthetic_params  # noqa: E402
from sanerf_hq_amd import _lib, raymarching as rm, synth  # noqa: E402
dev = torch.device("cuda:0")
out = {}
for hw in (800, 400):
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(hw, hw), hw, hw, device=dev)
    steps = [128, 64, 32]
    model = product_model(synthetic_params(steps, seed=0), steps, False, dev)
    plan = rm.RenderPlan(model, steps, torch.float16)
    base = None
    for name, kw in (("default", {}), ("bands2", dict(band_streams=2)), ("one_wg_bands2", dict(band_streams=2, experiment=_lib.EXP_FINAL_ONE_WG)),
                     ("one_wg_bands4", dict(band_streams=4, experiment=_lib.EXP_FINAL_ONE_WG)), ("one_wg_bands1", dict(band_streams=1, experiment=_lib.EXP_FINAL_ONE_WG)),
                     ("bands4", dict(band_streams=4))):
        tu = rm.Tuning(**kw)
        fn = lambda: rm.render_rays(plan, ro, rd, tile_w=hw, tuning=tu)      # noqa: E731
        ms = min(timeit(fn, 3, 20) for _ in range(3)) * 1e3
        img = fn()["image"].clone()
        base = img if base is None else base
        out.setdefault(f"{hw}x{hw}", {})[name] = {"ms": round(ms, 4), "bit_equal": bool(torch.equal(img, base))}
print(json.dumps(out))
