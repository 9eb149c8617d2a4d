"""Ray-tile sharding of one image across the GPUs of a node.

The path shards naturally: rays are independent and all state (hash tables,
MLP weights) is read-only at inference, so every rank holds a replica of the model, renders a
contiguous band of image rows, and ONE all-gather over RCCL/xGMI assembles the image
(rgb + depth + weights_sum = 5 floats per pixel; 1600x1600 -> 51 MB total, 6.4 MB per GPU).
There is no exchange inside the path.  The reference has no multi-GPU render at all (its DDP
branch is unreachable, nerf/trainer.py:119-122, and its only collectives gather evaluation
images, trainer.py:1578-1601).

One process per GPU, `torch.distributed` backend "nccl" (= RCCL on ROCm); the same code runs
under "gloo" on CPU tensors, which is how the sharding/gather logic is tested without GPUs.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

TILE_ROWS = 16   # the fused kernel's workgroup covers 16x16 pixels; bands start on tile boundaries


def shard_rows(H: int, world_size: int, rank: int, align: int = TILE_ROWS) -> Tuple[int, int]:
    """Rows [begin, end) of rank `rank`: contiguous bands of whole `align`-row tiles, the
    H % (align*world) remainder spread one tile at a time over the first ranks."""
    tiles = (H + align - 1) // align
    base, extra = divmod(tiles, world_size)
    t0 = rank * base + min(rank, extra)
    t1 = t0 + base + (1 if rank < extra else 0)
    return min(t0 * align, H), min(t1 * align, H)


def band_align(H: int, world_size: int) -> int:
    """Row alignment of the bands of an H-row image: whole 16-row workgroup tiles when that gives every rank the same
    number of rows, else whole 8-row wave tiles (1600 rows on 8 GPUs: 200 rows each -- the last workgroup row of a band
    is then half idle, but every rank launches the same grid and the all-gather needs no padding), else 16 (unequal
    bands, padded gather)."""
    for a in (TILE_ROWS, 8):
        if H % (a * world_size) == 0:
            return a
    return TILE_ROWS


def all_shards(H: int, world_size: int, align: int = TILE_ROWS) -> List[Tuple[int, int]]:
    return [shard_rows(H, world_size, r, align) for r in range(world_size)]


def gather_image(local: torch.Tensor, H: int, W: int, group: Optional[dist.ProcessGroup] = None,
                 align: int = TILE_ROWS, force_collective: bool = False) -> torch.Tensor:
    """local: [(rows_of_this_rank)*W, K] -> [H*W, K] on every rank (one all_gather, equal counts:
    bands are padded to the largest band).  force_collective: run the all-gather even in a one-rank group (the
    single-GPU self-test of the RCCL path, tools/rccl_selftest.py)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        return local
    rank = dist.get_rank(group)
    bands = all_shards(H, world, align)
    K = local.shape[-1]
    max_rows = max(e - b for b, e in bands)
    b, e = bands[rank]
    assert local.shape[0] == (e - b) * W, f"rank {rank}: expected {(e - b) * W} rays, got {local.shape[0]}"
    if all(e2 - b2 == max_rows for b2, e2 in bands):
        # equal bands (H a multiple of 16*world, e.g. the 800 x 800*N bench image): gather straight into the image
        out = local.new_empty(H * W, K)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    send = local.new_zeros(max_rows * W, K)
    send[: local.shape[0]] = local
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    return torch.cat([recv[r][: (bands[r][1] - bands[r][0]) * W] for r in range(world)], dim=0)


def render_image_sharded(render_rows: Callable[[int, int], torch.Tensor], H: int, W: int,
                         group: Optional[dist.ProcessGroup] = None, gather: bool = True, force_collective: bool = False) -> torch.Tensor:
    """render_rows(row_begin, row_end) -> [(row_end-row_begin)*W, K] for this rank's band.
    Returns the full [H*W, K] image on every rank (or the local band if gather=False)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    align = band_align(H, world)
    b, e = shard_rows(H, world, rank, align)
    local = render_rows(b, e) if e > b else None
    if local is None:
        raise RuntimeError(f"rank {rank} received an empty band: image height {H} has fewer than {world} row tiles")
    return gather_image(local, H, W, group, align, force_collective) if gather else local


def render_model_sharded(model, pose, intrinsics, H: int, W: int, group: Optional[dist.ProcessGroup] = None,
                         gather: bool = True, ray_fn: Optional[Callable] = None, force_collective: bool = False) -> torch.Tensor:
    """Whole-image render of a NeRFNetwork replica: [H*W, 5] = rgb | depth | weights_sum on every rank.
    Each rank generates the rays of its own band on its own device (nothing but the final image crosses xGMI).
    ray_fn(pose, intrinsics, H, W, device, row_begin, row_end) -> (rays_o, rays_d): defaults to the HIP
    generate_rays; the gloo tests pass a CPU twin."""
    if ray_fn is None:
        from .raymarching import generate_rays

        def ray_fn(pose, intrinsics, H, W, device, row_begin, row_end):
            return generate_rays(pose, intrinsics, H, W, device=device, row_begin=row_begin, row_end=row_end)

    device = next(model.parameters()).device

    def rows(b, e):
        rays_o, rays_d = ray_fn(pose, intrinsics, H, W, device, b, e)
        if device.type == "cuda":
            # the fused render writes rgb | depth | weights_sum straight into the all-gather payload (sn_render_io.out_stride): no concatenation
            band = torch.empty((e - b) * W, 5, device=device, dtype=torch.float32)
            out = model.render(rays_o, rays_d, staged=False, perturb=False, tile_w=W, packed=band)
            if out["image"].data_ptr() == band.data_ptr():
                return band
        else:
            out = model.render(rays_o, rays_d, staged=False, perturb=False, tile_w=W)
        return torch.cat([out["image"], out["depth"].unsqueeze(-1), out["weights_sum"].unsqueeze(-1)], dim=-1)   # (operator-chain routes, CPU twins of the tests)

    return render_image_sharded(rows, H, W, group, gather, force_collective)


class PipelinedGather:
    """All-gather of frame k overlapped with the render of frame k+1.

    `submit(local)` enqueues an asynchronous all-gather of this rank's band into one of `depth` rotating image
    buffers and returns immediately: the collective runs on the communicator's own stream (over xGMI) while the next
    frame's kernels run on the compute stream.  A buffer is reused only after its previous collective has finished.
    `drain()` waits for everything in flight and returns the most recent complete image.  Requires equal bands
    (H a multiple of 16*world); otherwise use `gather_image`."""

    def __init__(self, H: int, W: int, K: int, device, depth: int = 2, group: Optional[dist.ProcessGroup] = None,
                 dtype=torch.float32, align: Optional[int] = None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        bands = all_shards(H, self.world, band_align(H, self.world) if align is None else align)
        rows = bands[0][1] - bands[0][0]
        if any(e - b != rows for b, e in bands):
            raise ValueError(f"PipelinedGather needs equal bands: image height {H} is not a multiple of 8 * {self.world}")
        self.local_numel = rows * W
        self.images = [torch.empty(H * W, K, device=device, dtype=dtype) for _ in range(depth)]
        self.works = [None] * depth
        self.k = 0
        self.last = None
        self._waits: list = []          # (event before, event after) around every wait on a collective: what the compute stream lost to it
        self._timed = torch.device(device).type == "cuda"
        # How the band reaches the collective.  "in_place": the send buffer IS this rank's slice of the receive buffer (RCCL's in-place
        # all-gather: no staging copy).  "staged": the band is copied first -- gloo cannot alias, and an RCCL build that rejects the alias is
        # detected here, once, by a synchronous probe on a few bytes (the first real run on 8 GPUs must not die on it).
        self.path = "local" if not dist.is_initialized() else "staged (gloo cannot alias send and receive buffers)"
        if dist.is_initialized() and dist.get_backend(group) == "nccl":
            self.path = "in_place"
            try:
                probe = torch.zeros(self.world * 4, device=device, dtype=dtype)
                r = dist.get_rank(group)
                probe[r * 4:(r + 1) * 4] = float(r + 1)
                dist.all_gather_into_tensor(probe, probe[r * 4:(r + 1) * 4], group=group)
                want = torch.arange(1, self.world + 1, device=device, dtype=dtype).repeat_interleave(4)
                if not torch.equal(probe, want):
                    raise RuntimeError("in-place all_gather_into_tensor returned wrong data")
            except RuntimeError as e:       # noqa: BLE001
                self.path = f"staged (in-place all_gather_into_tensor rejected by the backend: {str(e).splitlines()[0][:160]})"

    def _wait(self, slot: int) -> None:
        w = self.works[slot]
        if w is None:
            return
        if self._timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            w.wait()
            b.record()
            self._waits.append((a, b))
        else:
            w.wait()
        self.works[slot] = None

    def band_buffer(self, rank: Optional[int] = None) -> torch.Tensor:
        """This rank's [rows * W, K] slice of the image buffer the NEXT submit() gathers into: render into it (render_rays(packed=...)) and
        hand it back to submit() -- the all-gather then runs in place, without a staging copy of the band."""
        slot = self.k % len(self.images)
        self._wait(slot)                            # the buffer's previous frame is complete
        r = (dist.get_rank(self.group) if dist.is_initialized() else 0) if rank is None else rank
        return self.images[slot][r * self.local_numel:(r + 1) * self.local_numel]

    def submit(self, local: torch.Tensor) -> None:
        assert local.shape[0] == self.local_numel, f"expected a band of {self.local_numel} rays, got {local.shape[0]}"
        slot = self.k % len(self.images)
        self._wait(slot)                            # the buffer's previous frame is complete
        if not dist.is_initialized():
            self.images[slot].copy_(local)
        else:
            src = local.contiguous()
            img = self.images[slot]
            aliased = img.data_ptr() <= src.data_ptr() < img.data_ptr() + img.numel() * img.element_size()
            if aliased and self.path != "in_place":
                src = src.clone()                   # band_buffer(): RCCL gathers in place (send = receive + rank * count); otherwise stage through a copy
            self.works[slot] = dist.all_gather_into_tensor(img, src, group=self.group, async_op=True)
        self.last = slot
        self.k += 1

    def drain(self) -> Optional[torch.Tensor]:
        for i in range(len(self.works)):
            self._wait(i)
        return None if self.last is None else self.images[self.last]

    def stats(self, reset: bool = True) -> dict:
        """{"path": how the band reaches the collective, "gather_wait_ms": total time the compute stream spent waiting on collectives since the
        last reset, "waits": their number}.  Synchronises the device (event timings)."""
        total = 0.0
        n = len(self._waits)
        if self._timed and n:
            torch.cuda.synchronize()
            total = sum(a.elapsed_time(b) for a, b in self._waits)
        if reset:
            self._waits = []
        return {"path": self.path, "gather_wait_ms": total, "waits": n}
