#!/usr/bin/env python3
"""sn_mlp_wide_backward_bits on k_mlp_wide_j<.., 2> vs k_mlp_wide<5> (experiments build: sn_debug_set("wide_bwd_j", v)) and the fp32-mask form k_mlp_wide<4>."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sanerf_hq_amd import _lib, ops, synth
gpu = torch.device("cuda:0")
N, din, n_out = 131072, 143, 16
ws = [torch.from_numpy(synth.linear_weight(o, i, 900 + k, 2.0)).to(gpu) for k, (o, i) in enumerate([(256, din), (256, 256), (n_out, 256)])]
x = torch.randn(N, din, device=gpu)
gy = torch.randn(N, n_out, device=gpu) * 10.0 ** torch.empty(N, 1, device=gpu).uniform_(-6, -1)
lib = _lib.lib()
def timed(label, bits, j):
    ops.WIDE_MLP_SIGN_BITS = bits
    if hasattr(lib, "sn_debug_set"):
        lib.sn_debug_set(b"wide_bwd_j", j)
    xs = x.clone().requires_grad_(True)
    wl = [w.clone().requires_grad_(False) for w in ws]     # data path only (no weight gradients)
    y = ops._wide_mlp_train.apply(xs, True, *wl)
    for _ in range(3):
        xs.grad = None; y.backward(gy, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        xs.grad = None; y.backward(gy, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) / 20:.3f} ms per backward data pass (incl. the weight pack)")
    return xs.grad.clone()
a = timed("k_mlp_wide<4> (fp32 masks)      ", False, 1)
b = timed("k_mlp_wide<5> (sign bits)        ", True, 0)
c = timed("k_mlp_wide_j<.., 2> (sign bits)  ", True, 1)
print("equal:", torch.equal(a, b), torch.equal(b, c))
