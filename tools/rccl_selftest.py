"""The RCCL leg of the multi-GPU path on ONE GPU: a one-rank "nccl" process group (= RCCL on ROCm), the band render of
dist.render_model_sharded with the all-gather forced, and the frame-pipelined PipelinedGather -- every gathered image must
equal the plain single-process render bit for bit.  What an N-GPU run does per rank, minus the other ranks (the reference's own
collectives: nerf/trainer.py:1578-1601, dist.all_gather of evaluation images).
usage (GPU box, repo root): python tools/rccl_selftest.py        -> prints 'rccl selftest OK ...' and exits 0"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

from sanerf_hq_amd import raymarching as rm, synth
from sanerf_hq_amd.dist import PipelinedGather, band_align, render_model_sharded, shard_rows

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
t0 = time.perf_counter()
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
ones = torch.ones(4, device=dev)
dist.all_reduce(ones)
assert float(ones.sum()) == 4.0
init_s = time.perf_counter() - t0
H = W = 256
steps = [128, 64, 32]
model = synth.product_model(synth.synthetic_params(steps, seed=3), steps, False, dev)
pose = synth.orbit_pose(1.0, 20.0, 30.0)
intr = synth.pinhole_intrinsics(H, W)
ro, rd = rm.generate_rays(pose, intr, H, W, device=dev)
with torch.no_grad():
    out = model.render(ro, rd, staged=False, perturb=False, tile_w=W)
    plain = torch.cat([out["image"], out["depth"].unsqueeze(-1), out["weights_sum"].unsqueeze(-1)], dim=-1).clone()
    gathered = render_model_sharded(model, pose, intr, H, W, force_collective=True)
    assert gathered.shape == plain.shape and torch.equal(gathered, plain), "all-gathered image differs from the plain render"
    align = band_align(H, 1)
    b, e = shard_rows(H, 1, 0, align)
    pipe = PipelinedGather(H, W, 5, dev, depth=2, align=align)
    for k in range(5):                           # frames in flight on RCCL's stream while the next one renders
        o = model.render(ro, rd, staged=False, perturb=False, tile_w=W)
        pipe.submit(torch.cat([o["image"], o["depth"].unsqueeze(-1), o["weights_sum"].unsqueeze(-1)], dim=-1))
    img = pipe.drain()
    torch.cuda.synchronize()
    assert torch.equal(img, plain), "pipelined all-gather differs from the plain render"
dist.barrier()
dist.destroy_process_group()
print(f"rccl selftest OK: backend nccl (RCCL), world 1, init+all_reduce {init_s:.2f} s, {H}x{W} image all-gathered (forced collective) "
      f"and frame-pipelined gather both bit-equal to the plain render")
