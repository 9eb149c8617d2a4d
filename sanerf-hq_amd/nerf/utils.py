"""Ray generation with the reference's `get_rays` contract (nerf/utils.py:182-304).

Full-image rays come from the HIP kernel (`raymarching.generate_rays`); pixel subsets
(`coords`, random pixels, random patches) are gathered from it.  The error-map /
incoherent-mask sampling branches belong to the reference's trainer-side data pipeline
 and raise NotImplementedError here.
"""
from __future__ import annotations

import numpy as np
import torch

from ..raymarching import generate_rays


def get_rays(poses, intrinsics, H, W, N=-1, patch_size=1, coords=None, device="cuda", incoherent_mask=None,
             include_incoherent_region=False, incoherent_mask_size=128, random_sample=False):
    """poses [1,4,4] cam2world, intrinsics [4] ndarray or [1,4] tensor -> dict(rays_o, rays_d[, i, j], inds_coarse)."""
    if torch.is_tensor(poses):
        device = poses.device if poses.is_cuda else device
    if isinstance(intrinsics, np.ndarray):
        intr = [float(v) for v in intrinsics.reshape(-1)[:4]]
    else:
        intr = [float(v) for v in intrinsics.reshape(-1, 4)[0].tolist()]
    pose = poses.reshape(-1, 4, 4)
    if pose.shape[0] != 1:
        raise NotImplementedError("get_rays: one camera per call (the reference's loaders use batch size 1)")
    rays_o, rays_d = generate_rays(pose[0], intr, H, W, device=device)
    results = {}
    if N > 0:
        if coords is not None:
            inds = (coords[:, 0] * W + coords[:, 1]).to(device).long()
        elif patch_size > 1 and not random_sample:
            if incoherent_mask is not None and include_incoherent_region:
                raise NotImplementedError("incoherent-mask patch sampling is trainer-side data logic (out of scope)")
            num_patch = N // (patch_size ** 2)
            ix = torch.randint(0, H - patch_size, size=[num_patch], device=device)
            iy = torch.randint(0, W - patch_size, size=[num_patch], device=device)
            base = torch.stack([ix, iy], dim=-1)
            pi, pj = torch.meshgrid(torch.arange(patch_size, device=device), torch.arange(patch_size, device=device), indexing="ij")
            offs = torch.stack([pi.reshape(-1), pj.reshape(-1)], dim=-1)
            pix = (base.unsqueeze(1) + offs.unsqueeze(0)).view(-1, 2)
            inds = pix[:, 0] * W + pix[:, 1]
        elif patch_size == 1 and not random_sample:
            raise NotImplementedError("error-map (incoherent_mask) pixel sampling is trainer-side data logic (out of scope)")
        else:
            inds = torch.randint(0, H * W, size=[N], device=device)
        rays_o, rays_d = rays_o[inds], rays_d[inds]
        results["i"] = inds % W
        results["j"] = torch.div(inds, W, rounding_mode="floor")
    else:
        inds = torch.arange(H * W, device=device)
    results["rays_o"] = rays_o
    results["rays_d"] = rays_d
    sx, sy = incoherent_mask_size / H, incoherent_mask_size / W
    cx = (torch.div(inds, W, rounding_mode="floor") * sx).long()
    cy = ((inds % W) * sy).long()
    results["inds_coarse"] = (cx * incoherent_mask_size + cy).long()
    return results
