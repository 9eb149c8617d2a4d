#!/usr/bin/env python3
"""Is sn_mlp_wide_forward_train deterministic, and are ALL of its elements right (not just the norm)?"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sanerf_hq_amd import _lib, synth
gpu = torch.device("cuda:0")
N, din, n_out, nl = 131072, 143, 2, 3
dims = [din, 256, 256, n_out]
ws = [torch.from_numpy(synth.linear_weight(dims[i + 1], dims[i], 900 + i, 2.0)).to(gpu) for i in range(nl)]
x = torch.randn(N, din, device=gpu)
desc = _lib.MlpDesc(); desc.num_layers, desc.activation, desc.skip_mask = nl, 1, 0; desc.dims[0] = din
for i, w in enumerate(ws):
    desc.weight[i], desc.bias[i], desc.dims[i + 1] = w.data_ptr(), None, w.shape[0]
def run():
    hs = [torch.full((N, 256), float("nan"), device=gpu) for _ in range(nl - 1)]
    y = torch.full((N, n_out), float("nan"), device=gpu)
    hid = (C.c_void_p * (nl - 1))(*[t.data_ptr() for t in hs])
    _lib.check(_lib.lib().sn_mlp_wide_forward_train(C.byref(desc), x.data_ptr(), N, hid, y.data_ptr(), _lib.stream()), "fwd")
    torch.cuda.synchronize()
    return hs + [y]
ref = run()
h = x.double(); refs = []
for i, w in enumerate(ws):
    h = torch.nn.functional.linear(h, w.double())
    if i + 1 < nl: h = torch.nn.functional.leaky_relu(h)
    refs.append(h)
for name, a, b in zip(("h0", "h1", "y"), ref, refs):
    d = (a.double() - b).abs()
    print(name, "max abs err", float(d.max()), "at", np.unravel_index(int(d.argmax()), d.shape), "scale", float(b.abs().mean()), "elements off by > 1e-4 scale:", int((d > 1e-4 * b.abs().mean()).sum()))
for it in range(5):
    out = run()
    print("run", it, [int((a != b).sum()) for a, b in zip(out, ref)], [np.unravel_index(int((a != b).double().argmax()), a.shape) if (a != b).any() else None for a, b in zip(out, ref)][:1])
