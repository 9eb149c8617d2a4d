#!/usr/bin/env python3
"""Distinct 128-byte lines per wave-wide gather instruction, per level (CPU only; round-4 verdict items 3 and 7).

What a gather instruction costs on the texture path follows the distinct lines it touches (tools/ubench/gathers.hip: 35 G wave-instr/s at <= 4
lines, 8.3 G at 32, 4.2 G at 64).  For the bench camera and the reference schedule's last stage this reproduces the table rows the lanes of one
instruction fetch -- for the main grid (F = 2: 4-byte rows with fp16 tables, 8 with fp32) under the 8x8-pixel wave tile of the render stages, and
for the F = 8 grids of the SAM / mask heads (32-byte rows) under the lane maps of k_feat_stage (8x8 pixels, one level), k_mlp_wide_j<3> (32
consecutive rays, one level) and k_mask16 (16 consecutive rays, two levels) -- and counts lines per (level, corner) instruction.  It also gives the
bounding box of a wave's cells per level: what a per-tile LDS patch (north_star's "LDS staging of per-tile grid voxels") would have to hold.
usage: python tools/gather_lines_table.py [--hw 400] [--tiles 32]"""
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


def rows_of(x01, r, size, off):
    """[n, T, 8] table rows of the 8 corners (gridencoder.cu:45-79, 145-149, 182)."""
    pos = np.clip(x01.astype(np.float32) * np.float32(r) - np.float32(0.5), 0, r - 1)
    pg = np.floor(pos).astype(np.uint32)
    dense = r ** 3 <= size
    out = []
    for c in range(8):
        qv = np.minimum(pg + np.array([c & 1, (c >> 1) & 1, (c >> 2) & 1], dtype=np.uint32), np.uint32(r - 1))
        row = (qv[..., 0] + qv[..., 1] * np.uint32(r) + qv[..., 2] * np.uint32(r * r)) if dense else (qv[..., 0] ^ (qv[..., 1] * P1) ^ (qv[..., 2] * P2))
        out.append((row % np.uint32(size)) + np.uint32(off))
    return np.stack(out, axis=-1), pg, dense


def distinct(a):
    """a [groups, lanes] -> mean number of distinct values per group."""
    s = np.sort(a, axis=1)
    return float((1 + (s[:, 1:] != s[:, :-1]).sum(axis=1)).mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=400)
    ap.add_argument("--tiles", type=int, default=32)
    a = ap.parse_args()
    steps = [128, 64, 32]
    H = W = a.hw
    params = synthetic_params(steps, seed=0)
    cfg = oracle_cfg(orc, params, steps)
    pose = synth.orbit_pose(1.0, 20.0, 30.0)
    fx, fy = synth.pinhole_intrinsics(H, W)[:2]
    ro, rd = orc.generate_rays(pose, fx, fy, W / 2.0, H / 2.0, H, W)
    rng = np.random.default_rng(0)
    ty = rng.integers(0, H // 8, a.tiles); tx = rng.integers(0, (W - 32) // 8, a.tiles)
    # per tile: an 8x8 pixel block (render stages, k_feat_stage) and the 32 consecutive pixels of its first row (the heads' ray-major tiles)
    idx88 = np.concatenate([((y * 8 + np.arange(8))[:, None] * W + (x * 8 + np.arange(8))[None, :]).ravel() for y, x in zip(ty, tx)])
    idx32 = np.concatenate([(y * 8) * W + x * 8 + np.arange(32) for y, x in zip(ty, tx)])
    bound = float(cfg.bound)

    def positions(idx):
        out = orc.render(cfg, ro[idx], rd[idx], debug=True)
        rb = out["real_bins2"].astype(np.float32)
        tmid = (rb[:, 1:] + rb[:, :-1]) / np.float32(2)
        p = ro[idx][:, None, :] + rd[idx][:, None, :] * tmid[..., None]
        z = orc.contract(p.reshape(-1, 3).astype(np.float32)).reshape(p.shape)
        return (z + bound) / (2 * bound)                                  # [n, T, 3]
    x88, x32 = positions(idx88), positions(idx32)
    T = x88.shape[1]
    n_w = a.tiles
    print(f"bench camera (orbit 1.0 / 20 / 30), {H}x{W}, last stage of [128,64,32] ({T} samples per ray), {n_w} wave tiles; lines = 128 bytes")
    for name, key, row_bytes_list in (("main grid (network.py:93: F = 2)", "grid", (4, 8)), ("SAM / mask grids (network.py:104: F = 8)", "s_grid", (32,))):
        g = GRIDS[key]
        offs, pls = orc.grid_layout(3, g["num_levels"], g["level_dim"], 2, 16, g["log2_hashmap_size"], g["desired_resolution"])
        L = g["num_levels"]
        res = orc.level_resolutions(L, float(np.log2(pls)), 16)
        print(f"\n{name}")
        if key == "grid":
            print(f"{'lvl':>3} {'res':>5} {'hash':>4} | 8x8 pixels x 1 level, lines per corner instruction: {'fp16 rows':>9} {'fp32 rows':>9} | cells of a wave: bbox <= 2^3  <= 4^3  <= 8^3 | patch rows (bbox, mean)")
        else:
            print(f"{'lvl':>3} {'res':>5} {'hash':>4} | lines per corner instruction: {'8x8 px (k_feat_stage)':>22} {'32 rays (k_mlp_wide_j<3>)':>26} {'16 rays (k_mask16, per level)':>30} | bbox(8x8) <= 2^3  <= 4^3  <= 8^3 | patch rows")
        per_lvl = []
        for l in range(L):
            r = int(res[l]); size = int(offs[l + 1] - offs[l])
            rows88, pg88, dense = rows_of(x88, r, size, int(offs[l]))
            cols = []
            for rbytes in row_bytes_list:
                per_line = 128 // rbytes
                ln = (rows88 // per_line).reshape(n_w, 64, T, 8).transpose(0, 2, 3, 1).reshape(-1, 64)
                cols.append(distinct(ln))
            if key != "grid":
                rows32, _, _ = rows_of(x32, r, size, int(offs[l]))
                ln32 = (rows32 // 4).reshape(n_w, 32, T, 8)
                cols.append(distinct(ln32.transpose(0, 2, 3, 1).reshape(-1, 32)))
                cols.append(distinct(ln32.reshape(n_w, 2, 16, T, 8).transpose(0, 1, 3, 4, 2).reshape(-1, 16)))
            pgw = pg88.reshape(n_w, 64, T, 3).astype(np.int64)
            ext = pgw.max(axis=1) - pgw.min(axis=1) + 2                   # vertices per axis of the wave's bounding box
            fit = [float((ext.max(axis=-1) <= m).mean()) for m in (2, 4, 8)]
            patch = float(ext.prod(axis=-1).mean())
            print(f"{l:>3} {r:>5} {'n' if dense else 'y':>4} | " + " ".join(f"{c:>{w}.1f}" for c, w in zip(cols, (9, 9) if key == "grid" else (22, 26, 30)))
                  + f" | {fit[0]:10.2f} {fit[1]:6.2f} {fit[2]:6.2f} | {patch:10.0f}")
            if key != "grid":
                # round 6 (row g1): line visits of one wave-sample of k_feat_stage at this level, three ways.  direct = the 8 corner instructions as
                # they are (lines merge inside an instruction only); unique = distinct lines over all 512 corner requests (what a perfect
                # de-duplicating stage could reach); patch = the wave's bounding box of vertices loaded row by row (what a per-wave LDS patch fetches:
                # on a hashed level every vertex is its own line)
                ln_all = (rows88 // 4).reshape(n_w, 64, T, 8).transpose(0, 2, 1, 3).reshape(-1, 512)
                per_lvl.append((l, r, dense, cols[0] * 8, distinct(ln_all), patch if not dense else patch / 4.0))
        if key != "grid":
            print("\nrow g1 (north_star: LDS staging of per-tile grid voxels), k_feat_stage, fp32 rows (32 B; fp16 rows: halve `patch` on dense levels only):")
            print(f"{'lvl':>3} {'res':>5} {'hash':>4} | line visits per wave-sample: {'direct (8 instr)':>16} {'unique lines':>13} {'bbox patch':>11} | patch / direct")
            for l, r, dense, d8, uq, pt in per_lvl:
                print(f"{l:>3} {r:>5} {'n' if dense else 'y':>4} | {'':29s}{d8:16.1f} {uq:13.1f} {pt:11.1f} | {pt / d8:8.2f}")
            tot = [sum(v[i] for v in per_lvl) for i in (3, 4, 5)]
            print(f"all {len(per_lvl)} levels: direct {tot[0]:.0f}, unique {tot[1]:.0f}, bbox patch {tot[2]:.0f} line visits per wave-sample.  A bounding-box patch fetches MORE lines than the "
                  f"direct gathers from level {next((v[0] for v in per_lvl if v[5] > v[3]), -1)} up: the samples of an 8x8-pixel tile at one depth index lie on a thin slanted sheet, "
                  "its box is mostly empty.")


if __name__ == "__main__":
    main()
