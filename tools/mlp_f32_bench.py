#!/usr/bin/env python3
"""Training forward of the mask head's MLP (131 072 x 143 -> 256 -> 256 -> 16): sn_mlp_wide_forward_train (one kernel, fp32 MFMA, fused
activations) against the layer-by-layer torch forward (hipBLASLt fp32 + one activation kernel per hidden layer).  usage (GPU box): python tools/mlp_f32_bench.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sanerf_hq_amd import _lib, synth
gpu = torch.device("cuda:0")
for N, din, n_out in ((131072, 143, 16), (131072, 143, 2), (524288, 143, 16), (65536, 64, 256)):
    dims = [din, 256, 256, n_out]
    ws = [torch.from_numpy(synth.linear_weight(dims[i + 1], dims[i], 900 + i, 2.0)).to(gpu) for i in range(3)]
    x = torch.randn(N, din, device=gpu)
    desc = _lib.MlpDesc(); desc.num_layers, desc.activation, desc.skip_mask = 3, 1, 0; desc.dims[0] = din
    for i, w in enumerate(ws):
        desc.weight[i], desc.bias[i], desc.dims[i + 1] = w.data_ptr(), None, w.shape[0]
    hs = [torch.empty(N, 256, device=gpu) for _ in range(2)]; y = torch.empty(N, n_out, device=gpu)
    hid = (C.c_void_p * 2)(*[t.data_ptr() for t in hs])
    def native():
        _lib.check(_lib.lib().sn_mlp_wide_forward_train(C.byref(desc), x.data_ptr(), N, hid, y.data_ptr(), _lib.stream()), "fwd")
    def blas():
        h = x
        for i, w in enumerate(ws):
            h = torch.nn.functional.linear(h, w)
            if i < 2: h = torch.nn.functional.leaky_relu(h, inplace=True)
        return h
    need = int(_lib.lib().sn_mlp_wide_workspace_bytes(C.byref(desc))); wsb = torch.empty(need, dtype=torch.uint8, device=gpu)
    def f16x3():
        _lib.check(_lib.lib().sn_mlp_wide_forward_train_f16x3(C.byref(desc), x.data_ptr(), N, hid, y.data_ptr(), wsb.data_ptr(), wsb.numel(), _lib.stream()), "fwd16")
    res = {}
    for name, fn in (("native", native), ("blas", blas), ("f16x3", f16x3)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20
    flops = 2.0 * N * sum(dims[i] * dims[i + 1] for i in range(3))
    print(f"N={N} {din}->256->256->{n_out}: native {res['native']:.3f} ms ({flops / res['native'] / 1e9:.1f} TFLOP/s fp32) | torch {res['blas']:.3f} ms ({flops / res['blas'] / 1e9:.1f}) | split-fp16 x3 with saved hidden outputs {res['f16x3']:.3f} ms")

lib = _lib.lib()
if os.environ.get("SN_TRACE"):
    import numpy as np
    buf = (C.c_ulonglong * 256)()
    lib.sn_mlp_wide_debug_trace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.sn_mlp_wide_debug_trace(buf, 64)
    t = np.array(buf[:40], dtype=np.int64)
    print("cycle stamps of the middle workgroup (last shape): start | x tile in LDS | chunk 0 parked | every chunk / epilogue:")
    print(" ".join(str(int(v)) for v in np.diff(t[:36])))
