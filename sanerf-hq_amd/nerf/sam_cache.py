"""SAM-feature cache container of the reference (`--feature_container cache`).

The reference stores the SAM encoder's feature map of every training view as `<workspace>/sam_cache/<img_name>.npy`,
float32 `[256, 64, 64]` (nerf/trainer.py:1069-1079, `store_sam_feautres`), and reads it back as the distillation target
or as the decoder's input (trainer.py:924-926: `np.load` -> tensor -> `unsqueeze(0)` = `[1, 256, 64, 64]`).  Same file
names, dtype and layout here, so caches written by either side are interchangeable.  The SAM encoder itself is outside
the hot path."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

SAM_FEATURE_SHAPE = (256, 64, 64)       # SAM ViT image-encoder output (trainer.py:520-523 comment: [1, 256, 64, 64])


class SamFeatureCache:
    def __init__(self, workspace: str, create: bool = True):
        self.path = os.path.join(workspace, "sam_cache")          # main.py / trainer.py: os.path.join(opt.workspace, 'sam_cache')
        if create:
            os.makedirs(self.path, exist_ok=True)

    def file_of(self, img_name: str) -> str:
        return os.path.join(self.path, img_name + ".npy")

    def __contains__(self, img_name: str) -> bool:
        return os.path.exists(self.file_of(img_name))

    def store(self, img_name: str, feature: torch.Tensor) -> str:
        """feature [256,64,64] or [1,256,64,64] -> <img_name>.npy (trainer.py:1076-1077)."""
        f = feature.detach()
        if f.dim() == 4 and f.shape[0] == 1:
            f = f[0]
        if f.dim() != 3:
            raise ValueError(f"SAM feature map must be [C,H,W] (or [1,C,H,W]), got {tuple(feature.shape)}")
        np.save(self.file_of(img_name), f.float().cpu().numpy())
        return self.file_of(img_name)

    def load(self, img_name: str, device=None) -> torch.Tensor:
        """-> [1,C,H,W] float32 on `device` (trainer.py:924-926)."""
        a = np.load(self.file_of(img_name))
        if a.dtype != np.float32 or a.ndim != 3:
            raise ValueError(f"{self.file_of(img_name)}: expected a float32 [C,H,W] array, found {a.dtype} {a.shape}")
        t = torch.from_numpy(a)
        return (t.to(device) if device is not None else t).unsqueeze(0)


def feature_map(samvit: torch.Tensor, h: int, w: int, size: Optional[tuple] = None) -> torch.Tensor:
    """Rendered per-ray features `results['samvit']` ([h*w, C] or [h, w, C]) -> `[1, C, h, w]`, bilinearly resized to `size`
    when given: the layout the reference compares with / stores as SAM features (trainer.py:536-542, 928-930)."""
    C = samvit.shape[-1]
    m = samvit.reshape(1, h, w, C).permute(0, 3, 1, 2).contiguous()
    if size is not None and tuple(size) != (h, w):
        m = torch.nn.functional.interpolate(m, size, mode="bilinear")
    return m
