#!/usr/bin/env python3
"""The library's general fp32 matrix product (sn_gemm_f32, csrc/linear.hip: the route of nn.Linear shapes no fused kernel covers) beside
torch.nn.functional.linear (rocBLAS) on layer shapes of heads that are not the reference's.  One JSON line: ms and fp32 TFLOP/s of both."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import timeit  # noqa: E402
from sanerf_hq_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
out = {}
for M, K, N in ((160000, 163, 128), (160000, 128, 128), (160000, 128, 256), (131072, 40, 128), (131072, 300, 300), (655360, 10, 16), (4096, 512, 512)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    t_own = min(timeit(lambda: ops.linear_forward(x, w, None, 1), 3, 10) for _ in range(3)) * 1e3
    t_lib = min(timeit(lambda: torch.relu_(torch.nn.functional.linear(x, w)), 3, 10) for _ in range(3)) * 1e3
    fl = 2.0 * M * K * N
    out[f"{M}x{K}->{N}"] = {"sn_gemm_f32_ms": round(t_own, 4), "sn_gemm_f32_tflops": round(fl / t_own / 1e9, 2), "torch_linear_relu_ms": round(t_lib, 4),
                           "torch_tflops": round(fl / t_lib / 1e9, 2), "max_abs_diff": float((ops.linear_forward(x, w, None, 1) - torch.relu_(torch.nn.functional.linear(x, w))).abs().max())}
print(json.dumps(out))
