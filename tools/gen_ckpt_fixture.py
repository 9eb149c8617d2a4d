#!/usr/bin/env python3
"""Checkpoint layout of the REFERENCE, captured by importing its own nerf/network.py in this container
(tests/golden/checkpoint_layout.json).  For each training mode of main.py (RGB field; --with_sam; --with_mask) it
records what `Trainer.save_checkpoint(full=True)` (nerf/trainer.py:1685-1718) puts under 'model' -- ordered keys, shapes,
dtypes of NeRFNetwork.state_dict() -- and under 'optimizer' -- the param-group structure Adam derives from
NeRFNetwork.get_params (network.py:206-230; main.py:283).  The 'stats' dictionary is built in trainer.py:151-157, a
module that cannot be imported here (imageio / wandb / torch_ema missing): its five keys are restated below with that
citation.  Run: python tools/gen_ckpt_fixture.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden as gg  # noqa: E402  (the reference import harness)


def main():
    rr, nn_, ru, enc = gg.install_reference()
    out = {"source": "reference nerf/network.py NeRFNetwork.state_dict() / get_params, imported in the build container",
           "top_level_keys_full": ["epoch", "global_step", "stats", "optimizer", "lr_scheduler", "scaler", "model"],   # trainer.py:1690-1705 (+ 'ema' when enabled)
           "top_level_keys_default": ["epoch", "global_step", "stats", "model"],
           "stats": {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None},               # trainer.py:151-157
           "modes": {}}
    for mode, kw in (("rgb", {}), ("sam", dict(with_sam=True)), ("mask", dict(with_mask=True)), ("sam+mask", dict(with_sam=True, with_mask=True))):
        opt = gg.make_opt(num_steps=[128, 64, 32], **kw)
        model = nn_.NeRFNetwork(opt)
        sd = model.state_dict()
        groups = model.get_params(1e-2)
        optim = torch.optim.Adam([dict(params=list(g["params"]), lr=g["lr"]) for g in groups], betas=(0.9, 0.99), eps=1e-15)   # main.py:283
        osd = optim.state_dict()
        out["modes"][mode] = {
            "state_dict": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()],
            "param_groups": [len(g["params"]) for g in osd["param_groups"]],
            "param_group_keys": sorted(k for k in osd["param_groups"][0] if k != "params"),
            "small_buffers": {k: v.tolist() for k, v in sd.items() if v.numel() <= 64 and not v.dtype.is_floating_point or k.startswith("aabb")},
        }
        print(mode, len(sd), "entries,", out["modes"][mode]["param_groups"])
    with open(os.path.join(ROOT, "tests", "golden", "checkpoint_layout.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote tests/golden/checkpoint_layout.json")


if __name__ == "__main__":
    main()
