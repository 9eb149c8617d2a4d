#!/usr/bin/env python3
"""Coherence model of the final stage's hash-grid gathers on the bench scene (CPU only).

For a sample of 8x8-pixel wave tiles it reproduces the table rows each lane fetches per
(sample, level, corner) instruction and counts (a) the distinct 128-byte lines per wave-wide
gather instruction and (b) the vertex bounding box of the wave per (sample, level).  With the
per-line cost measured by tools/ubench/gathers.hip this gives a lower bound for the gather
instruction stream; the bounding-box statistics size the per-wave LDS voxel cache idea.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402
from helpers import GRIDS, oracle_cfg, synthetic_params  # noqa: E402
from sanerf_hq_amd import synth  # noqa: E402

P1, P2 = np.uint32(2654435761), np.uint32(805459861)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--schedule", default="flat128")
    ap.add_argument("--hw", type=int, default=800)
    ap.add_argument("--tiles", type=int, default=48)
    ap.add_argument("--row-bytes", type=int, default=8)
    ap.add_argument("--xswap", type=int, default=1, help="model the half-wave x-pair exchange on hashed levels (0: plain corner order)")
    a = ap.parse_args()
    steps = [128] if a.schedule == "flat128" else [128, 64, 32]
    H = W = a.hw
    params = synthetic_params(steps, seed=0)
    cfg = oracle_cfg(orc, params, steps)
    pose = synth.orbit_pose(1.0, 20.0, 30.0)
    fx, fy = synth.pinhole_intrinsics(H, W)[:2]
    ro, rd = orc.generate_rays(pose, fx, fy, W / 2.0, H / 2.0, H, W)
    rng = np.random.default_rng(0)
    ty = rng.integers(0, H // 8, a.tiles); tx = rng.integers(0, W // 8, a.tiles)
    idx = np.concatenate([((y * 8 + np.arange(8))[:, None] * W + (x * 8 + np.arange(8))[None, :]).ravel() for y, x in zip(ty, tx)])
    out = orc.render(cfg, ro[idx], rd[idx], debug=True)
    k = len(steps) - 1
    rb = out[f"real_bins{k}"].astype(np.float32)
    T = rb.shape[1] - 1
    tmid = (rb[:, 1:] + rb[:, :-1]) / np.float32(2)
    p = ro[idx][:, None, :] + rd[idx][:, None, :] * tmid[..., None]
    z = orc.contract(p.reshape(-1, 3).astype(np.float32)).reshape(p.shape)
    bound = float(cfg.bound)
    x01 = (z + bound) / (2 * bound)                                    # [n, T, 3]
    g = GRIDS["grid"]
    offs, pls = orc.grid_layout(3, g["num_levels"], g["level_dim"], 2, 16, g["log2_hashmap_size"], g["desired_resolution"])
    L = g["num_levels"]
    res = orc.level_resolutions(L, float(np.log2(pls)), 16)
    rows_per_line = 128 // a.row_bytes
    n_w = a.tiles
    print(f"schedule {a.schedule}, {n_w} wave tiles x {T} samples, row {a.row_bytes} B")
    print(f"{'lvl':>3} {'res':>5} {'rows':>8} {'hash':>4} | {'lines/instr':>11} {'clk/instr':>9} | bbox<=2^3 <=3^3 <=4^3 | lines(bbox)")
    tot_clk = 0.0; tot_clk_cache = 0.0
    for l in range(L):
        r = res[l]; size = int(offs[l + 1] - offs[l])
        pos = np.clip(x01.astype(np.float32) * np.float32(r) - np.float32(0.5), 0, r - 1)
        pg = np.floor(pos).astype(np.uint32)                            # [n, T, 3]
        stride1 = r; dense = r ** 3 <= size                              # gridencoder.cu:45-79 (stride = resolution)
        lines = []
        for c in range(8):
            q = np.minimum(pg + np.array([c & 1, (c >> 1) & 1, (c >> 2) & 1], dtype=np.uint32), np.uint32(r - 1))   # gridencoder.cu:182
            if dense:
                row = q[..., 0] + q[..., 1] * np.uint32(stride1) + q[..., 2] * np.uint32(stride1 * stride1)
            else:
                row = q[..., 0] ^ (q[..., 1] * P1) ^ (q[..., 2] * P2)
            row = (row % np.uint32(size)) + np.uint32(offs[l])
            line = (row // rows_per_line).reshape(n_w, 64, T)
            lines.append(line)
        if a.xswap and not dense:
            # half-wave x-pair exchange (render.hip, XSWAP): instruction 2p serves corners 2p and 2p+1 of lanes 0-31,
            # instruction 2p+1 those of lanes 32-63
            ex = []
            for pr in range(4):
                lo = np.concatenate([lines[2 * pr][:, :32], lines[2 * pr + 1][:, :32]], axis=1)
                hi = np.concatenate([lines[2 * pr][:, 32:], lines[2 * pr + 1][:, 32:]], axis=1)
                ex += [lo, hi]
            lines = ex
        lines = np.stack([np.array([[len(np.unique(ln[w, :, j])) for j in range(T)] for w in range(n_w)]) for ln in lines])   # [8, n_w, T]
        clk = np.maximum(17.5, 2.3 * lines)
        pgw = pg.reshape(n_w, 64, T, 3).astype(np.int64)
        ext = pgw.max(axis=1) - pgw.min(axis=1) + 2                      # vertices per axis  [n_w, T, 3]
        fit = [(ext.max(axis=-1) <= m).mean() for m in (2, 3, 4)]
        # one cooperative gather of the 4^3 block: distinct lines of its 64 vertices
        tot_clk += clk.sum(axis=0).mean()
        fits4 = ext.max(axis=-1) <= 4
        cache_clk = np.where(fits4, 17.5 * 1 + 8 * 4.0, clk.sum(axis=0))
        tot_clk_cache += cache_clk.mean() if l < 9 else clk.sum(axis=0).mean()
        print(f"{l:>3} {r:>5} {size:>8} {'n' if dense else 'y':>4} | {lines.mean():>11.1f} {clk.mean():>9.1f} | {fit[0]:7.2f} {fit[1]:5.2f} {fit[2]:5.2f} |")
    n_rays = H * W
    wave_samples = n_rays / 64 * T
    ms = lambda c: c * wave_samples / 256 / 2.4e9 * 1e3
    print(f"gather-stream bound: {tot_clk:.0f} clk per wave-sample -> {ms(tot_clk):.2f} ms for {H}x{W};  "
          f"with a per-wave 4^3 vertex cache on levels 0-8: {tot_clk_cache:.0f} clk -> {ms(tot_clk_cache):.2f} ms")


if __name__ == "__main__":
    main()
