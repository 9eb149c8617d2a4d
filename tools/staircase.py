#!/usr/bin/env python3
"""Kernel time of the fused stages as a function of the NUMBER OF WORKGROUPS of a launch (fp16 tables): a 1600-pixel-wide image
of 16*k rows is 100*k workgroups of 16x16 pixels on 512 resident slots (2 per CU).  Shows the round quantisation an 8-way row-band
split of BASELINE configs[3] runs into (200 rows -> 1300 workgroups = 2.54 rounds) and what a tail policy could recover.
usage (GPU box): python tools/staircase.py [W]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from helpers import product_model, synthetic_params  # noqa: E402
from sanerf_hq_amd import _lib, raymarching as rm, synth  # noqa: E402

dev = torch.device("cuda:0")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
H = 1600
lib = _lib.lib()
ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, H), H, H, device=dev)
ro = ro.view(H, H, 3)[:, :W].contiguous()
rd = rd.view(H, H, 3)[:, :W].contiguous()
tiles_x = (W + 15) // 16
ks = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 21, 24, 25, 26, 28, 31, 36, 41, 46, 51, 52]
for steps in ([128, 64, 32], [128]):
    model = product_model(synthetic_params(steps, seed=0), steps, False, dev)
    plan = rm.RenderPlan(model, steps, torch.float16)
    print(f"schedule {steps}, W={W}: rows  workgroups  rounds(512)  | pack prop0 prop1 final ms | us per final round-equivalent")
    for k in ks:
        rows = 16 * k
        if rows > H:
            break
        o = ro[:rows].reshape(-1, 3).contiguous()
        d = rd[:rows].reshape(-1, 3).contiguous()
        for _ in range(3):
            rm.render_rays(plan, o, d, tile_w=W)
        torch.cuda.synchronize()
        lib.sn_rm_profile_enable(1)
        n = 10
        for _ in range(n):
            rm.render_rays(plan, o, d, tile_w=W)
        torch.cuda.synchronize()
        ms = (C.c_float * 8)()
        cnt = (C.c_int32 * 8)()
        _lib.check(lib.sn_rm_profile_read(ms, cnt, 8), "profile_read")
        lib.sn_rm_profile_enable(0)
        nwg = tiles_x * k
        per = [ms[i] / n for i in range(6)]
        print(f"  {rows:5d} {nwg:6d} {nwg / 512:6.2f} | {per[0]:.3f} {per[1]:.3f} {per[2]:.3f} {per[4]:.3f} | {per[4] * 1e3 / (nwg / 512):.1f}")
