#!/usr/bin/env python3
"""Phase trace of the binned grid backward (k_bin_scatter / k_bin_accum) at the mask-field step's size: a library built with
-DSN_BIN_TRACE=1 (VARIANT_FILE=grid_binned tools/build_variant.sh bintrace -DSN_BIN_TRACE=1) records the shader clock of 16 workgroups
per kernel at every phase boundary.  usage (GPU box): SN_LIB=ab/bintrace.so python tools/bin_trace.py [C] [rays] [samples]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle as orc  # noqa: E402
from sanerf_hq_amd import _lib, ops  # noqa: E402
from sanerf_hq_amd.gridencoder import grid_encode  # noqa: E402

Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
offs, pls = orc.grid_layout(3, 16, Cc, 2, 16, 19, 512 if Cc == 8 else 4096)
o = rng.uniform(0.1, 0.9, (R, 1, 3)); d = rng.normal(size=(R, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
t = np.sort(rng.uniform(-0.5, 0.5, (R, T, 1)), axis=1)
x = np.clip(o + d * t, 0.0, 1.0).reshape(-1, 3).astype(np.float32)
B = x.shape[0]
xt = torch.from_numpy(x).to(dev)
emb = torch.zeros(int(offs[-1]), Cc, device=dev).uniform_(-1e-4, 1e-4).requires_grad_(True)
offt = torch.from_numpy(np.asarray(offs, dtype=np.int32)).to(dev)
g = torch.randn(B, 16 * Cc, device=dev)
ops.GRID_BACKWARD_MODE = "binned"
out = grid_encode(xt, emb, offt, pls, 16, False, 0, False, 0)


def bwd():
    emb.grad = None
    out.backward(g, retain_graph=True)


for _ in range(3):
    bwd()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10):
    bwd()
ev[1].record()
torch.cuda.synchronize()
print(f"C={Cc} B={B}: backward {ev[0].elapsed_time(ev[1]) / 10:.3f} ms (incl. zero fill of the table)")
lib = _lib.lib()
if hasattr(lib, "sn_bin_debug_trace"):
    buf = (C.c_ulonglong * 512)()
    lib.sn_bin_debug_trace.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.sn_bin_debug_trace(buf)
    tr = np.array(buf[:], dtype=np.int64).reshape(2, 16, 16)
    names = (["scatter: zero hist + rank (rows, LDS atomics)", "barrier", "cursor atomics + barrier", "stores drained"],
             ["accum: loads issued, counters zeroed", "count (waits for the loads)", "barrier", "scan", "place", "barrier", "sum + store drained"])
    for k in range(2):
        print("kernel", "k_bin_scatter" if k == 0 else "k_bin_accum", "(cycles per phase, 16 workgroups spread over the launch)")
        n = len(names[k])
        for wg in range(16):
            st = tr[k, wg, :n + 1]
            if st[0] == 0:
                continue
            dl = np.diff(st)
            print(f"  wg#{wg:2d} total {st[n] - st[0]:7d} | " + " ".join(f"{v:6d}" for v in dl))
        print("   phases: " + " | ".join(names[k]))
else:
    print("(library without SN_BIN_TRACE: no phase trace)")
