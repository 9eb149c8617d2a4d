#!/usr/bin/env python3
"""RGB-mode and mask-mode training steps alone, for rocprofv3 kernel traces (tools/bench_configs.py holds the timings)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import timeit  # noqa: E402
from helpers import make_opt, synthetic_params  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402
from sanerf_hq_amd.nerf import NeRFNetwork  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "rgb"
H = W = 512; N = 4096
roF, rdF = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 60.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
pix = torch.from_numpy((synth.hash_u01(N, 99) * (H * W)).astype(np.int64)).to(dev)
ro, rd = roF[pix].contiguous(), rdF[pix].contiguous()
if mode in ("rgb", "rgb_noprop"):
    opt = make_opt(); opt.lambda_proposal, opt.lambda_distort = 1.0, 0.0
    model = NeRFNetwork(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_params([128, 64, 32], seed=1).items()}, strict=False)
    model = model.to(dev).train()
    gt = torch.from_numpy(synth.hash_uniform((N, 3), 42, 0.0, 1.0)).to(dev)
    from sanerf_hq_amd.optim import Adam as HipAdam
    optim = (torch.optim.Adam if os.environ.get("SN_PROFILE_TORCH_ADAM") else HipAdam)(model.get_params(1e-2), eps=1e-15)
    upd = mode == "rgb"                                        # "rgb_noprop": the step the reference runs 4 times of 5 after step 3000 (trainer.py:372-373)
    def step():
        optim.zero_grad(set_to_none=True)
        o = model.render(ro, rd, staged=False, bg_color=1, perturb=True, update_proposal=upd)
        loss = torch.nn.functional.mse_loss(o["image"], gt)
        if upd:
            loss = loss + o["proposal_loss"]
        loss.backward()
        optim.step()
else:
    opt = make_opt(with_mask=True)
    model = NeRFNetwork(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_params([128, 64, 32], heads=True, seed=1).items()}, strict=False)
    model = model.to(dev).train()
    for n_, p in model.named_parameters():
        p.requires_grad_(n_.startswith("m_grid") or n_.startswith("mask_mlp"))
    labels = torch.from_numpy((synth.hash_u01(N, 100) < 0.5).astype(np.int64)).to(dev)
    from sanerf_hq_amd.optim import Adam as HipAdam
    optim = (torch.optim.Adam if os.environ.get("SN_PROFILE_TORCH_ADAM") else HipAdam)([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=1e-15)
    def step():
        optim.zero_grad(set_to_none=True)
        o = model.render(ro, rd, staged=False, bg_color=1, perturb=False, update_proposal=False, return_mask=1)
        rm.mask_nll(o["instance_mask_logits"], labels, 1e-6).mean().backward()     # trainer.py:419-428 as one kernel
        optim.step()
print(mode, "step ms", timeit(step, 3, 10) * 1e3)
