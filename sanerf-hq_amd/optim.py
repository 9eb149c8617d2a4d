"""Adam with the update of each parameter tensor as ONE HIP kernel (csrc/optim.hip: sn_adam_step).

Drop-in for the optimiser the reference constructs (main.py:283: `torch.optim.Adam(model.get_params(lr), eps=1e-15)`):
same constructor arguments, same `state_dict` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), same dense update
rule -- every element moves each step by its decaying momentum, touched by a sample or not.  torch's default foreach
implementation walks the 160 MiB state of a head grid in ~20 multi-tensor kernels; this is one pass per tensor.
CUDA fp32 contiguous parameters only; anything else raises (no silent fallback).

`capturable=True` keeps the step count on the device (like torch.optim.Adam(capturable=True)): no host value is baked into the launch, so a whole
training step can be captured in a HIP graph and replayed (sanerf_hq_amd.graph.GraphedStep).

`lazy=True` (per parameter group, opt-in) switches that group to a touched-elements-only update: an element whose
gradient is exactly zero in a step is skipped altogether (moments do not decay, the parameter does not coast on its momentum) --
torch.optim.SparseAdam's semantics with the non-zeros of the dense gradient as the sparse pattern.  Not the reference's
optimiser: meant for the hash tables, of which a 4096-ray batch touches a few percent of the rows.
"""
from __future__ import annotations

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False, lazy=False,
                 capturable=False):
        if amsgrad:
            raise ValueError("sanerf_hq_amd.optim.Adam: amsgrad is not implemented (the reference does not use it)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize, lazy=bool(lazy),
                                      capturable=bool(capturable)))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if g.is_sparse or not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("sanerf_hq_amd.optim.Adam handles dense contiguous fp32 CUDA parameters only")
                if not g.is_contiguous():
                    g = g.contiguous()
                st = self.state[p]
                cap = bool(group.get("capturable", False))
                if len(st) == 0:
                    # host counter, as torch's non-capturable Adam keeps it; capturable: a device float that the kernel reads when it RUNS
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device) if cap else torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1                                                   # (capturable: a one-element device kernel, captured with the rest)
                lr = group["lr"]
                _lib.check(lib.sn_adam_step(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                                            float(lr), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                            0 if cap else int(st["step"].item()), st["step"].data_ptr() if cap else None,
                                            int(bool(group["maximize"])), _lib.ADAM_LAZY if group.get("lazy", False) else 0, _lib.stream()), "sn_adam_step")
                # the kernel wrote through the raw pointer: tell autograd / version-keyed caches (RenderPlan.check_range's fp16
                # range guard, memoised host copies) that the tensor changed, as an in-place torch op would have
                torch.autograd.graph.increment_version(p)
        return loss
