"""sn_linear_wgrad against torch's dy^T @ x on the mask-head shapes of a 4096-ray training step (131072 samples).
Run on the GPU box: python tools/wgrad_bench.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sanerf_hq_amd import _lib  # noqa: E402


def timeit(fn, warm=3, iters=20):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3     # us


def main():
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    out = {}
    for M, K, N in ((131072, 256, 256), (131072, 143, 256), (131072, 256, 2), (131072, 419, 256), (524288, 32, 64), (524288, 10, 16), (524288, 16, 1), (131072, 64, 64)):
        x, dy = torch.randn(M, K, device=dev), torch.randn(M, N, device=dev)
        ws = torch.empty(int(lib.sn_linear_wgrad_workspace_bytes(M, K, N)), dtype=torch.uint8, device=dev)
        dw = torch.empty(N, K, device=dev)

        def ours():
            _lib.check(lib.sn_linear_wgrad(x.data_ptr(), dy.data_ptr(), M, K, N, dw.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "wgrad")
        t_o, t_t = timeit(ours), timeit(lambda: dy.t() @ x)
        out[f"M{M}_K{K}_N{N}"] = {"sn_linear_wgrad_us": round(t_o, 1), "torch_us": round(t_t, 1),
                                   "tflops": round(2.0 * M * K * N / t_o * 1e-6, 1), "operand_GBps": round(4.0 * M * (K + N) / t_o * 1e-3, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
