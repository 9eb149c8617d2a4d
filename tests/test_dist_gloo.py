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


# ---- the real render_model_sharded signature (model.render(rays_o, rays_d, staged=False, perturb=False, tile_w=W)) ----
class _StubField(torch.nn.Module):
    """Stands in for NeRFNetwork on CPU: `render` is a deterministic function of each ray alone, like the real path."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([[0.3, -0.2, 0.9], [0.5, 0.1, -0.4], [-0.7, 0.6, 0.2]]))
        self.calls = []

    def render(self, rays_o, rays_d, staged=False, perturb=False, tile_w=0, **kw):
        assert staged is False and perturb is False and tile_w > 0 and rays_o.shape == rays_d.shape
        assert rays_o.shape[0] % tile_w == 0, "a band is whole image rows"
        self.calls.append(rays_o.shape[0])
        with torch.no_grad():
            rgb = torch.sigmoid(rays_d @ self.w.T + rays_o)
            return {"image": rgb, "depth": rays_d.norm(dim=-1), "weights_sum": torch.ones(rays_d.shape[0])}


def _cpu_rays(pose, intr, H, W, device, b, e):
    """nerf/utils.py:269-287 for rows [b, e) in torch (what sn_rm_generate_rays computes on the GPU)."""
    fx, fy, cx, cy = intr
    pose = torch.as_tensor(np.asarray(pose, dtype=np.float32))
    j, i = torch.meshgrid(torch.arange(b, e, dtype=torch.float32) + 0.5, torch.arange(W, dtype=torch.float32) + 0.5, indexing="ij")
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], dim=-1).reshape(-1, 3)
    rays_d = dirs @ pose[:3, :3].T
    return pose[:3, 3].expand_as(rays_d).contiguous(), rays_d.contiguous()


def _model_worker(rank, world, port, H, W, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sanerf_hq_amd import synth
    from sanerf_hq_amd.dist import band_align, render_model_sharded, shard_rows
    model = _StubField()
    pose, intr = synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W)
    img = render_model_sharded(model, pose, intr, H, W, ray_fn=_cpu_rays)
    b, e = shard_rows(H, world, rank, band_align(H, world))
    assert model.calls == [(e - b) * W], "each rank renders exactly its own band, once"
    torch.save(img, os.path.join(out_dir, f"m{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W", [(64, 40), (1600 // 8, 24), (72, 16)])   # equal 16-row bands, equal 8-row-aligned bands (the 8-GPU config-4 split), unequal bands
def test_render_model_sharded_assembles_the_single_process_image(tmp_path, H, W):
    world = 2
    mp.spawn(_model_worker, args=(world, _free_port(), H, W, str(tmp_path)), nprocs=world, join=True)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sanerf_hq_amd import synth
    model = _StubField()
    ro, rd = _cpu_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, "cpu", 0, H)
    out = model.render(ro, rd, tile_w=W)
    full = torch.cat([out["image"], out["depth"].unsqueeze(-1), out["weights_sum"].unsqueeze(-1)], dim=-1)
    for r in range(world):
        assert torch.equal(torch.load(os.path.join(tmp_path, f"m{r}.pt")), full), f"rank {r}: gathered image differs"


def test_band_alignment_of_the_config4_image():
    """1600 rows: 16-row tiles for 1/2/4 ranks, 8-row wave tiles at 8 ranks (200 rows each, equal bands, no padding)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sanerf_hq_amd.dist import all_shards, band_align
    for world, want_align in ((1, 16), (2, 16), (4, 16), (8, 8)):
        a = band_align(1600, world)
        assert a == want_align
        bands = all_shards(1600, world, a)
        assert bands[0][0] == 0 and bands[-1][1] == 1600 and all(e - b == 1600 // world for b, e in bands)
        assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))


def _pipe8_worker(rank, world, port, H, W, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sanerf_hq_amd.dist import PipelinedGather, band_align, shard_rows
    align = band_align(H, world)
    b, e = shard_rows(H, world, rank, align)
    pg = PipelinedGather(H, W, 5, "cpu", depth=2, align=align)
    for k in range(3):                                   # bench.py's step(): render INTO this rank's slice of the image buffer, gather in place
        band = pg.band_buffer()
        assert band.shape == ((e - b) * W, 5) and band.data_ptr() == pg.images[k % 2][b * W:e * W].data_ptr()
        band.copy_(_fake_render(H, W)(b, e) + float(k))
        pg.submit(band)
    img = pg.drain().clone()
    torch.save({"img": img, "band": (b, e), "align": align}, os.path.join(out_dir, f"q{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_config4_split_gathers_in_place(tmp_path):
    """World size 8 (the round-4 verdict: the gloo tests stopped at 2 ranks): band_align(1600, 8) = 8-row wave tiles, 200 rows per rank,
    PipelinedGather through band_buffer() -- the band is written into the image buffer and gathered without a staging copy, as bench.py's
    N > 1 branch does -- three frames through two rotating buffers; every rank ends with the whole last frame."""
    world, H, W = 8, 1600, 4
    mp.spawn(_pipe8_worker, args=(world, _free_port(), H, W, str(tmp_path)), nprocs=world, join=True)
    full = _fake_render(H, W)(0, H) + 2.0
    bands = []
    for r in range(world):
        d = torch.load(os.path.join(tmp_path, f"q{r}.pt"))
        assert d["align"] == 8 and torch.equal(d["img"], full), f"rank {r}"
        bands.append(d["band"])
    assert bands == [(200 * r, 200 * (r + 1)) for r in range(world)]
