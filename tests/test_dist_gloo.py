"""CPU, world_size 2 over gloo: the ray-tile shard + all-gather of dist.py assembles exactly the
single-process image (the N>1 path of bench.py uses the same functions over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_render(H, W):
    def rows(b, e):   # deterministic "render": value depends only on the pixel
        n = torch.arange(b * W, e * W, dtype=torch.float32)
        return torch.stack([n, n * 0.5, -n, n % 7, torch.ones_like(n)], dim=-1)
    return rows


def _worker(rank, world, port, H, W, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sanerf_hq_amd.dist import render_image_sharded, shard_rows
    img = render_image_sharded(_fake_render(H, W), H, W)
    b, e = shard_rows(H, world, rank)
    torch.save({"img": img, "band": (b, e)}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W", [(64, 48), (80, 33), (1000, 8)])
def test_sharded_render_equals_single_process(tmp_path, H, W):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, H, W, str(tmp_path)), nprocs=world, join=True)
    full = _fake_render(H, W)(0, H)
    bands = []
    for r in range(world):
        d = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        assert torch.equal(d["img"], full), f"rank {r} did not receive the full image"
        bands.append(d["band"])
    assert bands[0][0] == 0 and bands[0][1] == bands[1][0] and bands[1][1] == H


def _pipe_worker(rank, world, port, H, W, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sanerf_hq_amd.dist import PipelinedGather, shard_rows
    b, e = shard_rows(H, world, rank)
    pg = PipelinedGather(H, W, 5, "cpu", depth=2)
    frames = []
    for k in range(5):                                   # five frames through two rotating buffers
        pg.submit(_fake_render(H, W)(b, e) + float(k))
    img = pg.drain().clone()
    torch.save(img, os.path.join(out_dir, f"p{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_overlaps_frames_and_keeps_order(tmp_path):
    """bench.py's N>1 path: asynchronous all-gathers into rotating buffers; the drained image is the LAST frame."""
    world, H, W = 2, 64, 24
    mp.spawn(_pipe_worker, args=(world, _free_port(), H, W, str(tmp_path)), nprocs=world, join=True)
    full = _fake_render(H, W)(0, H) + 4.0
    for r in range(world):
        assert torch.equal(torch.load(os.path.join(tmp_path, f"p{r}.pt")), full)
