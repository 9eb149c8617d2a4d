#!/usr/bin/env python3
"""Gradient error of the mask-field step against the reference fixture (tests/golden/train_c5.npz) for three forwards of the mask MLP:
BLAS fp32, the fp32-MFMA kernel, split-fp16 x3 products (emulated layer by layer through sn_mlp_wide_forward)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
from helpers import golden, params_from_spec, spec_of, make_opt
from sanerf_hq_amd import _lib, ops
from sanerf_hq_amd.nerf import NeRFNetwork
gpu = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)

def layer_f16x3(x, w):
    L = _lib.lib()
    desc = _lib.MlpDesc(); desc.num_layers, desc.activation, desc.skip_mask = 1, 1, 0
    desc.dims[0], desc.dims[1] = x.shape[1], w.shape[0]; desc.weight[0], desc.bias[0] = w.data_ptr(), None
    need = int(L.sn_mlp_wide_workspace_bytes(C.byref(desc)))
    ws = torch.empty(need, dtype=torch.uint8, device=gpu)
    out = torch.empty(x.shape[0], w.shape[0], device=gpu)
    _lib.check(L.sn_mlp_wide_forward(C.byref(desc), None, None, 0.0, x.data_ptr(), x.shape[0], out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "fwd")
    return out

orig_forward = ops._wide_mlp_train.forward
def fwd_f16x3(ctx, x, leaky, *weights):
    hs, h = [], x.reshape(-1, x.shape[-1]).contiguous()
    for i, w in enumerate(weights):
        h = layer_f16x3(h, w.contiguous())
        if i + 1 < len(weights):
            h = torch.nn.functional.leaky_relu(h, inplace=True) if leaky else torch.relu_(h)
            hs.append(h.reshape(*x.shape[:-1], 256))
    h = h.reshape(*x.shape[:-1], h.shape[-1])
    ctx.save_for_backward(x, *hs, *weights)
    ctx.nl, ctx.leaky = len(weights), bool(leaky)
    return h

g = golden("train_c5")
params = params_from_spec(spec_of(g))
for mode in ("blas", "f32mfma", "f16x3"):
    ops.WIDE_MLP_FORWARD_NATIVE = mode == "f32mfma"
    ops.WIDE_MLP_FORWARD_F16X3 = False           # (the f16x3 row goes through the layer-by-layer emulation below; tools/mlp_f32_bench.py times the real kernel)
    ops._wide_mlp_train.forward = staticmethod(fwd_f16x3) if mode == "f16x3" else orig_forward
    model = NeRFNetwork(make_opt(with_mask=True))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    model = model.to(gpu).train()
    for n_, p in model.named_parameters():
        p.requires_grad_(n_.startswith("m_grid") or n_.startswith("mask_mlp"))
    out = model.render(T(g["rays_o"]), T(g["rays_d"]), staged=False, bg_color=1, perturb=False, update_proposal=False, return_rgb=0, return_feats=0, return_mask=1)
    logits = out["instance_mask_logits"]
    eps = float(g["epsilon"])
    pm = torch.softmax(logits, dim=-1).clamp(min=eps, max=1 - eps)
    loss = (-torch.log(torch.gather(pm, -1, T(g["labels"])[..., None]))).mean()
    loss.backward()
    def rel(got, ref):
        got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
        return np.linalg.norm(got - ref) / np.linalg.norm(ref)
    errs = [rel(lin.weight.grad.cpu().numpy(), g[f"mask_mlp_grad{i}"]) for i, lin in enumerate(model.mask_mlp[0].net)]
    ge = model.m_grid.embeddings.grad
    errs.append(rel(ge[T(g["m_grid_rows"])].cpu().numpy(), g["m_grid_grad_rows"]))
    print(f"{mode:8s} logits max err {float(np.abs(logits.detach().cpu().numpy() - g['logits']).max()):.2e}  rel-L2 of grads: W0 {errs[0]:.2e} W1 {errs[1]:.2e} W2 {errs[2]:.2e} table rows {errs[3]:.2e}   (bar 1e-3)")
