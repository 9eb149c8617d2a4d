"""Two PROCESSES sharing cuda:0, rendezvous over "gloo" with DEVICE tensors: what an N > 1 run does that the one-rank RCCL self-test
cannot show on a one-GPU box -- several ranks joining one group, each rendering its own row band with the real kernels, the bands of the
others arriving through a collective on device tensors, PipelinedGather with two frames in flight.  (RCCL refuses two ranks on one
device; gloo stages device tensors through the host, so this exercises the sharding / gather LOGIC with real HIP tensors, not xGMI.)
Every rank checks the assembled image against its own single-process render bit for bit.
usage (GPU box, repo root): python tools/two_rank_device_selftest.py      -> prints 'two-rank selftest OK ...' and exits 0"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    import torch.distributed as dist
    from sanerf_hq_amd import raymarching as rm, synth
    from sanerf_hq_amd.dist import PipelinedGather, band_align, render_model_sharded, shard_rows
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.full((4,), float(rank + 1), device=dev)
    dist.all_reduce(t)                                          # device tensor through the group: every rank joined
    assert float(t[0]) == world * (world + 1) / 2
    steps = [128, 64, 32]
    model = synth.product_model(synth.synthetic_params(steps, seed=3), steps, False, dev)
    pose = synth.orbit_pose(1.0, 20.0, 30.0)
    for H, W in ((256, 256), (400, 120), (72, 48)):            # equal 16-row bands, equal 8-row-aligned bands (200 rows each), unequal bands (padded gather)
        intr = synth.pinhole_intrinsics(H, W)
        ro, rd = rm.generate_rays(pose, intr, H, W, device=dev)
        with torch.no_grad():
            out = model.render(ro, rd, staged=False, perturb=False, tile_w=W)
            plain = torch.cat([out["image"], out["depth"].unsqueeze(-1), out["weights_sum"].unsqueeze(-1)], dim=-1).clone()
            gathered = render_model_sharded(model, pose, intr, H, W)
        assert gathered.is_cuda and gathered.shape == plain.shape and torch.equal(gathered, plain), f"rank {rank}: {H}x{W} gathered image differs"
        align = band_align(H, world)
        b, e = shard_rows(H, world, rank, align)
        if all(shard_rows(H, world, r, align)[1] - shard_rows(H, world, r, align)[0] == e - b for r in range(world)):
            pipe = PipelinedGather(H, W, 5, dev, depth=2, align=align)
            rob, rdb = rm.generate_rays(pose, intr, H, W, device=dev, row_begin=b, row_end=e)
            with torch.no_grad():
                for _ in range(4):
                    o = model.render(rob, rdb, staged=False, perturb=False, tile_w=W)
                    pipe.submit(torch.cat([o["image"], o["depth"].unsqueeze(-1), o["weights_sum"].unsqueeze(-1)], dim=-1))
            img = pipe.drain()
            torch.cuda.synchronize()
            assert torch.equal(img, plain), f"rank {rank}: {H}x{W} pipelined gather differs"
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} of {world} OK")


if __name__ == "__main__":
    if "RANK" in os.environ:
        worker()
        sys.exit(0)
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    ok = True
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
            err += "\nTIMEOUT"
        ok = ok and p.returncode == 0 and f"rank {r} of {world} OK" in out
        if p.returncode != 0:
            print(f"---- rank {r} rc={p.returncode}\n{out[-1500:]}\n{err[-3000:]}")
    if ok:
        print(f"two-rank selftest OK: {world} processes on cuda:0, gloo with device tensors, render_model_sharded + PipelinedGather bit-equal to the single-process render on every rank")
    sys.exit(0 if ok else 1)
