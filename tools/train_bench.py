#!/usr/bin/env python3
"""Timings of the two training steps alone (the same numbers tools/bench_configs.py reports, without its other configurations):
  rgb   RGB-mode step (trainer.py:360-392): 4096 rays, [128,64,32], everything trainable, MSE + proposal loss
  mask  BASELINE configs[4]: mask-field step, 4096 rays, field frozen
usage: train_bench.py [rgb|mask|both]  -> one JSON line.  Eager forward+backward, + single-pass Adam, the step as a HIP graph, and (rgb) the
step without proposal update (4 of 5 steps after step 3000, trainer.py:372-373)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import timeit  # noqa: E402
from helpers import make_opt, synthetic_params  # noqa: E402
from sanerf_hq_amd import ops, raymarching as rm, synth  # noqa: E402
from sanerf_hq_amd.graph import GraphedStep  # noqa: E402
from sanerf_hq_amd.nerf import NeRFNetwork  # noqa: E402
from sanerf_hq_amd.optim import Adam as HipAdam  # noqa: E402

dev = torch.device("cuda:0")
H = W = 512
N = 4096
best = lambda fn, n=3: min(timeit(fn) for _ in range(n))        # noqa: E731


def rays():
    roF, rdF = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 60.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
    pix = torch.from_numpy((synth.hash_u01(N, 99) * (H * W)).astype(np.int64)).to(dev)
    return roF[pix].contiguous(), rdF[pix].contiguous()


def rgb():
    ro, rd = rays()
    opt = make_opt()
    opt.lambda_proposal, opt.lambda_distort = 1.0, 0.0
    model = NeRFNetwork(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_params([128, 64, 32], seed=1).items()}, strict=False)
    model = model.to(dev).train()
    gt = torch.from_numpy(synth.hash_uniform((N, 3), 42, 0.0, 1.0)).to(dev)
    box = {"optim": HipAdam(model.get_params(1e-2), eps=1e-15), "upd": True}

    def fwd_bwd():
        box["optim"].zero_grad(set_to_none=True)
        o = model.render(ro, rd, staged=False, bg_color=1, perturb=True, update_proposal=box["upd"])
        loss = torch.nn.functional.mse_loss(o["image"], gt)
        if box["upd"]:
            loss = loss + o["proposal_loss"]
        loss.backward()

    def step():
        fwd_bwd()
        box["optim"].step()
    out = {"fwd_bwd_ms": round(best(fwd_bwd) * 1e3, 3), "step_ms": round(best(step) * 1e3, 3)}
    box["upd"] = False
    out["fwd_bwd_without_proposal_update_ms"] = round(best(fwd_bwd) * 1e3, 3)
    out["step_without_proposal_update_ms"] = round(best(step) * 1e3, 3)
    for upd, key in ((True, "step_as_hip_graph_ms"), (False, "step_without_proposal_update_as_hip_graph_ms")):
        box["upd"] = upd
        box["optim"] = HipAdam(model.get_params(1e-2), eps=1e-15, capturable=True)
        try:
            g = GraphedStep(step, warmup=3)
            out[key] = round(best(g) * 1e3, 3)
            del g
        except Exception as e:   # noqa: BLE001
            out[key] = f"failed: {type(e).__name__}: {e}"
    return out


def mask():
    ro, rd = rays()
    opt = make_opt(with_mask=True)
    model = NeRFNetwork(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_params([128, 64, 32], heads=True, seed=1).items()}, strict=False)
    model = model.to(dev).train()
    for n_, p in model.named_parameters():
        p.requires_grad_(n_.startswith("m_grid") or n_.startswith("mask_mlp"))
    labels = torch.from_numpy((synth.hash_u01(N, 100) < 0.5).astype(np.int64)).to(dev)
    train = [p for p in model.parameters() if p.requires_grad]
    box = {"optim": HipAdam(train, lr=1e-3, eps=1e-15)}
    ops.WGRAD_SIDE_STREAM = True

    def fwd_bwd():
        box["optim"].zero_grad(set_to_none=True)
        o = model.render(ro, rd, staged=False, bg_color=1, perturb=False, update_proposal=False, return_mask=1)
        rm.mask_nll(o["instance_mask_logits"], labels, 1e-6).mean().backward()

    def step():
        fwd_bwd()
        box["optim"].step()
    out = {"fwd_bwd_ms": round(best(fwd_bwd) * 1e3, 3), "step_ms": round(best(step) * 1e3, 3)}
    box["optim"] = HipAdam(train, lr=1e-3, eps=1e-15, capturable=True)
    try:
        g = GraphedStep(step, warmup=3)
        out["step_as_hip_graph_ms"] = round(best(g) * 1e3, 3)
        del g
    except Exception as e:   # noqa: BLE001
        out["step_as_hip_graph_ms"] = f"failed: {type(e).__name__}: {e}"
    ops.WGRAD_SIDE_STREAM = False
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    res = {}
    if which in ("rgb", "both"):
        res["rgb_training_step_4096_rays"] = rgb()
    if which in ("mask", "both"):
        res["c5_mask_training_step_4096_rays"] = mask()
    print(json.dumps(res))
