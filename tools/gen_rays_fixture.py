#!/usr/bin/env python3
"""tests/golden/rays_multi.npz: the reference's own nerf/utils.get_rays (imported in the build container) for the
per-ray-camera call of provider.py:908-913 (`random_image_batch`: poses = self.poses[index], intrinsics likewise, one
image index per ray) -- inputs and outputs.  Run: python tools/gen_rays_fixture.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden as gg  # noqa: E402


def main():
    rr, nn_, ru, enc = gg.install_reference()
    rng = np.random.default_rng(77)
    H, W, n_img, N = 96, 128, 7, 300
    cams = np.stack([gg.synth.orbit_pose(0.8 + 0.1 * k, 10.0 + 7 * k, 50.0 * k) for k in range(n_img)]).astype(np.float32)
    intr = np.stack([np.array(gg.synth.pinhole_intrinsics(H, W, 45.0 + 3 * k), np.float32) for k in range(n_img)])
    index = rng.integers(0, n_img, N)
    coords = np.stack([rng.integers(0, H, N), rng.integers(0, W, N)], axis=-1).astype(np.int64)
    res = ru.get_rays(torch.from_numpy(cams[index]), torch.from_numpy(intr[index]), H, W, N, coords=torch.from_numpy(coords),
                      device="cpu", incoherent_mask_size=32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rays_multi.npz"), cams=cams, intr=intr, index=index, coords=coords,
                        HW=np.array([H, W]), rays_o=res["rays_o"].numpy(), rays_d=res["rays_d"].numpy(), i=res["i"].numpy(),
                        j=res["j"].numpy(), inds_coarse=res["inds_coarse"].numpy())
    print("wrote tests/golden/rays_multi.npz", res["rays_d"].shape)


if __name__ == "__main__":
    main()
