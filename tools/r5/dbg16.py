import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from sanerf_hq_amd import _lib, raymarching as rm
from sanerf_hq_amd.gridencoder import GridEncoder
from sanerf_hq_amd.nerf.network import SkipConnMLP
gpu = torch.device("cuda:0")
for (N, T_, n_inst, L, E) in [(300, 32, 2, 16, 15), (4000, 32, 2, 16, 15)]:
    torch.manual_seed(N + L)
    enc = GridEncoder(input_dim=3, num_levels=L, level_dim=8, base_resolution=16, log2_hashmap_size=15, desired_resolution=512).to(gpu)
    with torch.no_grad():
        enc.embeddings.uniform_(-1.0, 1.0)
    mlp = SkipConnMLP(L * 8 + E, n_inst, 256, 3, skip_layers=[], bias=False).to(gpu)
    xyz = torch.rand(N, T_, 3, device=gpu) * 2.2 - 1.1
    extra = torch.randn(N, T_, E, device=gpu)
    w = torch.rand(N, T_, device=gpu)
    a = rm.mask_head(w, xyz, extra, enc, mlp, 1.0).clone()
    _lib.check(_lib.lib().sn_debug_set(b"mask_head16", 8), "debug_set")
    for i in range(6):
        b = rm.mask_head(w, xyz, extra, enc, mlp, 1.0).clone()
        d = (a - b).abs().max(dim=-1).values
        bad = torch.nonzero(d > 1e-5).flatten()
        print(N, "run", i, "max", float(d.max()), "bad rays", bad.numel(), bad[:24].tolist(), "groups of 16:", sorted(set((bad // 16).tolist()))[:12])
    _lib.check(_lib.lib().sn_debug_set(b"mask_head16", 0), "debug_set")
