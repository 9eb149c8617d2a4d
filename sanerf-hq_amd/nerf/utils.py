"""Ray generation with the reference's `get_rays` contract (nerf/utils.py:182-304).

Full-image rays come from the HIP kernel `raymarching.generate_rays`; pixel subsets (`coords`, random pixels, random
patches, and the error-map / incoherent-mask draws of utils.py:214-259) are drawn on the device with torch ops and
turned into rays by `raymarching.rays_from_pixels`, with one camera per ray if asked -- no host round trip and no
H x W intermediate in a training step.
"""
from __future__ import annotations

import numpy as np
import torch

from ..raymarching import generate_rays


def get_rays(poses, intrinsics, H, W, N=-1, patch_size=1, coords=None, device="cuda", incoherent_mask=None,
             include_incoherent_region=False, incoherent_mask_size=128, random_sample=False):
    """poses [1 or N,4,4] cam2world, intrinsics [4] ndarray or [1 or N,4] tensor -> dict(rays_o, rays_d[, i, j], inds_coarse).

    N <= 0: every pixel of one camera (HIP generate_rays).  N > 0: the reference's four ways of drawing N pixels
    (utils.py:209-263: given `coords`, random patches, error-map / incoherent-mask multinomial, uniform) on the device,
    and rays of exactly those pixels from `rays_from_pixels` -- with one camera per ray when `poses` has N entries
    (provider.py:908-913, `random_image_batch`).  The full H x W image is never generated for a subset."""
    from ..raymarching import rays_from_pixels
    if torch.is_tensor(poses):
        device = poses.device if poses.is_cuda else device
    pose = (poses if torch.is_tensor(poses) else torch.as_tensor(np.asarray(poses, dtype=np.float32))).reshape(-1, 4, 4)
    n_cam = pose.shape[0]
    if isinstance(intrinsics, np.ndarray):
        intr_t = torch.as_tensor(intrinsics.reshape(-1)[:4].astype(np.float32)).reshape(1, 4)
    else:
        intr_t = intrinsics.reshape(-1, 4).float()
    results = {}
    if N > 0:
        if coords is not None:
            inds = (coords[:, 0] * W + coords[:, 1]).to(device).long()
        elif patch_size > 1 and not random_sample:
            if incoherent_mask is not None and include_incoherent_region:      # utils.py:214-225: one patch around a drawn coarse cell
                centre = torch.multinomial(incoherent_mask.to(device=device, dtype=torch.float32).reshape(1, -1), 1).reshape(-1)
                cxm = torch.div(centre, incoherent_mask_size, rounding_mode="floor")
                cym = centre % incoherent_mask_size
                ix = torch.clamp(cxm * (H / incoherent_mask_size) - patch_size // 2, min=0, max=H - patch_size - 1).long()
                iy = torch.clamp(cym * (W / incoherent_mask_size) - patch_size // 2, min=0, max=W - patch_size - 1).long()
            else:
                num_patch = N // (patch_size ** 2)
                ix = torch.randint(0, H - patch_size, size=[num_patch], device=device)
                iy = torch.randint(0, W - patch_size, size=[num_patch], device=device)
            base = torch.stack([ix, iy], dim=-1)
            pi, pj = torch.meshgrid(torch.arange(patch_size, device=device), torch.arange(patch_size, device=device), indexing="ij")
            offs = torch.stack([pi.reshape(-1), pj.reshape(-1)], dim=-1)
            pix = (base.unsqueeze(1) + offs.unsqueeze(0)).view(-1, 2)
            inds = pix[:, 0] * W + pix[:, 1]
        elif patch_size == 1 and not random_sample:                            # utils.py:245-257: error-map draw without replacement
            if incoherent_mask is None:
                raise RuntimeError("get_rays: patch_size=1 without random_sample draws from incoherent_mask (utils.py:246); none given")
            m = incoherent_mask.to(device=device, dtype=torch.float32).reshape(1, -1)
            inds_coarse = torch.multinomial(m, N, replacement=False)               # [1, N] in [0, size*size)
            gx = torch.div(inds_coarse, incoherent_mask_size, rounding_mode="floor")
            gy = inds_coarse % incoherent_mask_size
            sx, sy = H / incoherent_mask_size, W / incoherent_mask_size
            px = (gx * sx + torch.rand(1, N, device=device) * sx).long().clamp(max=H - 1)
            py = (gy * sy + torch.rand(1, N, device=device) * sy).long().clamp(max=W - 1)
            inds = (px * W + py)[0]
            results["inds_coarse"] = inds_coarse                                   # the caller updates its error map with these
        else:
            inds = torch.randint(0, H * W, size=[N], device=device)
        n_rays = inds.shape[0]
        if n_cam not in (1, n_rays) or intr_t.shape[0] not in (1, n_rays):
            raise RuntimeError(f"get_rays: {n_cam} poses / {intr_t.shape[0]} intrinsics for {n_rays} rays (each must be 1 or one per ray)")
        rays_o, rays_d = rays_from_pixels(pose.to(device), intr_t.to(device), inds, W)
        results["i"] = inds % W
        results["j"] = torch.div(inds, W, rounding_mode="floor")
    else:
        if n_cam != 1:
            raise RuntimeError("get_rays: a full image needs exactly one camera")
        intr = [float(v) for v in intr_t.reshape(-1)[:4].tolist()]
        rays_o, rays_d = generate_rays(pose[0], intr, H, W, device=device)
        inds = torch.arange(H * W, device=device)
    results["rays_o"] = rays_o
    results["rays_d"] = rays_d
    if results.get("inds_coarse") is None:                                      # utils.py:294-300
        sx, sy = incoherent_mask_size / H, incoherent_mask_size / W
        cx = (torch.div(inds, W, rounding_mode="floor") * sx).long()
        cy = ((inds % W) * sy).long()
        results["inds_coarse"] = (cx * incoherent_mask_size + cy).long()
    return results


def collate_rays(poses, intrinsics, H, W, num_rays, index=None, images=None, masks=None, error_map=None, cam_near_far=None,
                 random_image_batch=True, use_error_map=False, error_map_size=128, num_local_sample=0, local_patch_size=1):
    """Device-side core of a training step's `NeRFDataset.collate` (nerf/provider.py:894-1114, RGB / mask modes): choose
    the cameras, draw the pixels, build their rays and gather the supervision at exactly those pixels -- all on the GPU the
    dataset was preloaded to, no H x W intermediate, no host round trip.

      poses [M,4,4], intrinsics [M,4] (or [1,4] / ndarray[4]), images [M,H,W,3|4] uint8, masks [M,H,W,C], error_map
      [M, S*S], cam_near_far [M,2] -- the preloaded dataset tensors; `index` = the loader's image index (a [1] tensor or
      int) used when random_image_batch is off (provider.py:908-913 draws one camera per ray otherwise).
      use_error_map: pixels from the error-map multinomial (provider.py:959-963) instead of uniformly (:964-968).
      num_local_sample > 0: the mixed-sampling extra of provider.py:970-984 -- that many patches of local_patch_size^2
      rays around cells drawn from their images' error maps, appended to rays / masks / error_maps / cam_near_far.

    Returns the reference's result keys: rays_o, rays_d, index, poses, intrinsics, H, W, inds_coarse[, images, masks,
    error_maps, cam_near_far], plus the pixel coordinates i, j."""
    dev = poses.device
    M = poses.shape[0]
    if isinstance(intrinsics, np.ndarray):
        intrinsics = torch.as_tensor(intrinsics.reshape(-1, 4).astype(np.float32), device=dev)
    intr_all = intrinsics.reshape(-1, 4).float()
    if random_image_batch:
        index = torch.randint(0, M, size=(num_rays,), device=dev)                        # provider.py:912
        random_sample = True
    else:
        index = torch.as_tensor([index] if isinstance(index, int) else index, device=dev).reshape(-1).long()
        random_sample = False
    cam_pose = poses[index]
    cam_intr = intr_all[index] if intr_all.shape[0] == M else intr_all
    emap = None if error_map is None else error_map[index]
    if use_error_map:
        rays = get_rays(cam_pose, cam_intr, H, W, num_rays, device=dev, patch_size=1, incoherent_mask=emap,
                        include_incoherent_region=True, incoherent_mask_size=error_map_size, random_sample=random_sample)
    else:
        rays = get_rays(cam_pose, cam_intr, H, W, num_rays, device=dev, patch_size=1, incoherent_mask=None,
                        include_incoherent_region=False, incoherent_mask_size=H, random_sample=True)
    res = {"H": H, "W": W, "index": index, "poses": cam_pose, "intrinsics": cam_intr, "rays_o": rays["rays_o"], "rays_d": rays["rays_d"],
           "i": rays["i"], "j": rays["j"], "inds_coarse": rays["inds_coarse"]}
    loc = None
    if num_local_sample > 0:                                                             # provider.py:970-984
        li = torch.randint(0, M, size=(num_local_sample,), device=dev)
        li_exp = li[:, None].expand(-1, local_patch_size * local_patch_size).reshape(-1)
        parts = []
        for k in range(num_local_sample):    # one patch per drawn image (the reference's single call draws one centre for all)
            intr_k = intr_all[li[k:k + 1]] if intr_all.shape[0] == M else intr_all[:1]      # the patch's own camera model
            parts.append(get_rays(poses[li[k:k + 1]].expand(local_patch_size * local_patch_size, 4, 4), intr_k, H, W, 1, device=dev,
                                  patch_size=local_patch_size, incoherent_mask=None if error_map is None else error_map[li[k:k + 1]],
                                  include_incoherent_region=error_map is not None, incoherent_mask_size=error_map_size, random_sample=False))
        loc = {k_: torch.cat([p_[k_] for p_ in parts], 0) for k_ in ("rays_o", "rays_d", "i", "j")}
        res["poses"] = torch.cat([res["poses"], poses[li_exp]], 0)
        res["rays_o"] = torch.cat([res["rays_o"], loc["rays_o"]], 0)
        res["rays_d"] = torch.cat([res["rays_d"], loc["rays_d"]], 0)
    if images is not None:                                                               # provider.py:1004-1014
        res["images"] = images[index, rays["j"], rays["i"]].float() / 255
    if masks is not None:                                                                # provider.py:1020-1035
        m = masks[index, rays["j"], rays["i"]]
        if loc is not None:
            m = torch.cat([m, masks[li_exp, loc["j"], loc["i"]]], 0)
        res["masks"] = m.view(-1, masks.shape[-1])
    if error_map is not None:                                                            # provider.py:1038-1060
        # the reference scales both pixel coordinates by error_map_size / H (its datasets are square); the column uses W here,
        # which is the same number for a square image and stays inside the map otherwise
        sj, si = error_map_size / H, error_map_size / W
        e = error_map[index, (rays["j"] * sj).long() * error_map_size + (rays["i"] * si).long()]
        if loc is not None:
            e = torch.cat([e, error_map[li_exp, (loc["j"] * sj).long() * error_map_size + (loc["i"] * si).long()]], 0)
        res["error_maps"] = e.view(-1)
    else:
        res["error_maps"] = None
    if cam_near_far is not None:                                                         # provider.py:1063-1068
        c = cam_near_far[index]
        if loc is not None:
            c = torch.cat([c, cam_near_far[li_exp]], 0)
        res["cam_near_far"] = c
    return res


# ---------------------------------------------------------------------------------------------
# checkpoints in the reference's format (nerf/trainer.py:1685-1741 save, :1779-1800 load)
# ---------------------------------------------------------------------------------------------
def save_checkpoint(model, path, epoch=0, global_step=0, stats=None, optimizer=None, lr_scheduler=None):
    """{'epoch', 'global_step', 'stats', 'model': state_dict[, 'optimizer', 'lr_scheduler']} -- what the reference's
    Trainer.save_checkpoint writes (the `full=True` extras only when given), so its load_checkpoint reads it back."""
    import torch
    state = {"epoch": int(epoch), "global_step": int(global_step),
             "stats": stats if stats is not None else {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None},
             "model": model.state_dict()}
    if optimizer is not None:
        state["optimizer"] = optimizer.state_dict()
    if lr_scheduler is not None:
        state["lr_scheduler"] = lr_scheduler.state_dict()
    torch.save(state, path)
    return state


def load_checkpoint(model, checkpoint, map_location=None):
    """Load a reference checkpoint (file path or already-loaded dict).  Like trainer.py:1779-1800: a dict without a
    'model' entry is taken as a bare state_dict (strict); otherwise state['model'] is loaded with strict=False and
    (missing, unexpected, state) is returned so that the caller can restore epoch / optimizer."""
    import torch
    state = torch.load(checkpoint, map_location=map_location, weights_only=False) if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__") else checkpoint
    if "model" not in state:
        model.load_state_dict(state)
        return [], [], {}
    missing, unexpected = model.load_state_dict(state["model"], strict=False)
    return list(missing), list(unexpected), state


def freeze_loaded_parameters(model, model_dict):
    """main.py:249-256: after loading a pretrained radiance field into a SAM / mask model, every parameter that came
    from the checkpoint is frozen; the new heads stay trainable."""
    frozen = []
    for k, v in model.named_parameters():
        if k in model_dict:
            v.requires_grad = False
            frozen.append(k)
    return frozen
