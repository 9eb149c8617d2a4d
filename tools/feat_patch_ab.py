#!/usr/bin/env python3
"""A/B of the feature stage's per-wave LDS patch on dense levels (sn_render_tuning.feat_patch): BASELINE configs[2]
(400x400 + SAM head) and 800x800, fp32 and fp16 tables; f_feat bit-equal; k_feat_stage time from the library's event profile."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import timeit  # noqa: E402
from helpers import make_opt, synthetic_params  # noqa: E402
from sanerf_hq_amd import _lib, raymarching as rm, synth  # noqa: E402
from sanerf_hq_amd.nerf import NeRFNetwork  # noqa: E402

dev = torch.device("cuda:0")
model = NeRFNetwork(make_opt(with_sam=True))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_params([128, 64, 32], heads=True, seed=1).items()}, strict=False)
model = model.to(dev).eval()
lib = _lib.lib()
out = {}
for hw in (400, 800):
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(hw, hw), hw, hw, device=dev)
    for dt, tag in ((torch.float32, "f32"), (torch.float16, "f16")):
        plan = rm.RenderPlan(model, [128, 64, 32], dt, feat_encoder=model.s_grid)
        res = {}
        for name, fp, lg in (("patch", 1, 0), ("direct", 0, 0), ("patch_lg1", 1, 1), ("direct_lg1", 0, 1)):
            tu = rm.Tuning(feat_patch=fp, feat_levels=lg)
            fn = lambda: rm.render_rays(plan, ro, rd, tile_w=hw, tuning=tu, out=res.setdefault(name + "_buf", {}))      # noqa: E731
            fn(); torch.cuda.synchronize()
            lib.sn_rm_profile_enable(1)
            ms_total = min(timeit(fn, 2, 10) for _ in range(3)) * 1e3
            lib.sn_rm_profile_enable(0)
            lib.sn_rm_profile_enable(1)
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            ms = (C.c_float * 8)(); cnt = (C.c_int32 * 8)()
            _lib.check(lib.sn_rm_profile_read(ms, cnt, 8), "profile_read")
            lib.sn_rm_profile_enable(0)
            res[name] = {"render_ms": round(ms_total, 4), "kernel_class_ms_per_frame": [round(ms[i] / 10, 4) if cnt[i] else None for i in range(8)]}
            res[name + "_feat"] = fn()["f_feat"].clone()
        out[f"{hw}x{hw}_{tag}"] = {k: res[k] for k in ("patch", "direct", "patch_lg1", "direct_lg1")}
        out[f"{hw}x{hw}_{tag}"]["f_feat_bit_equal"] = bool(torch.equal(res["patch_feat"], res["direct_feat"]) and torch.equal(res["patch_lg1_feat"], res["direct_feat"]))
print(json.dumps(out))
