"""A/B of the two proposal-stage kernels on linear-order ray batches of growing size (picks the default of Tuning.prop_sp_max_rays).
Run on the GPU box: python tools/prop_sp_ab.py [N]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import product_model, synthetic_params  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402


def timeit(fn, warm=3, iters=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = torch.device("cuda:0")
    steps = [int(t) for t in os.environ.get("SN_AB_STEPS", "128,64,32").split(",")]
    model = product_model(synthetic_params(steps, seed=1), steps, False, dev)
    H = W = 512
    roF, rdF = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 60.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
    out = {}
    for dt in (torch.float32, torch.float16):
        plan = rm.RenderPlan(model, steps, dt)
        for N in ([int(sys.argv[1])] if len(sys.argv) > 1 else (1024, 4096, 8192, 16384, 32768, 65536, 131072)):
            pix = torch.from_numpy((synth.hash_u01(N, 99) * (H * W)).astype(np.int64)).to(dev)
            ro, rd = roF[pix].contiguous(), rdF[pix].contiguous()
            row = {}
            for name, v, vf in (("lane", "0", "0"), ("sp_prop", "100000000", "0"), ("sp", "100000000", "100000000")):
                rm.tuning.prop_sp_max_rays = int(v) if int(v) else -1
                rm.tuning.final_sp_max_rays = int(vf) if int(vf) else -1
                row[name + "_ms"] = round(timeit(lambda: rm.render_rays(plan, ro, rd, tile_w=0)) * 1e3, 4)
            out[f"{str(dt).split('.')[-1]}_N{N}"] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
