"""Dataset-side voxelisation on the GPU: the host work of
SemanticKITTISCN.__getitem__ (data/semantic_kitti/semantic_kitti_dataloader.py:216-238) --
`augment_and_scale_3d` without augmentation (data/utils/augmentation_3d.py:41-44), the int64 cast,
the in-range mask and the `sparse_quantize` dedupe -- on device tensors, reusing libftx's sorted-unique
kernel.  Bit-exact with the reference's numpy path (tests/golden/voxel_coords.npz)."""
from __future__ import annotations

import torch

from .. import functional as spf


def points_to_voxels(points: torch.Tensor, scale: int = 20, full_scale: int = 4096):
    """points (N,3) float32 on the GPU -> (coords (M,3) int64 of the kept, deduped voxels,
    keep_index (M,) int64 rows of `points` they come from, in sorted-key order like np.unique)."""
    if not points.is_cuda or points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("points_to_voxels: expected a (N,3) float32 CUDA tensor")
    coords = points * float(scale)                       # float32 multiply, as numpy does
    coords = coords - coords.min(0).values               # translate to the positive octant
    ci = coords.to(torch.int64)                          # astype(np.int64): truncation
    valid = (ci.min(1).values >= 0) & (ci.max(1).values < full_scale)
    rows = torch.nonzero(valid).squeeze(1)               # dataloader side: a host sync here is fine
    cv = ci[rows]
    key = ((cv[:, 0] * full_scale + cv[:, 1]) * full_scale + cv[:, 2]).contiguous()
    uniq, first, cnt = spf.unique_sorted(key)            # ascending key, first occurrence of each
    n = int(cnt.item())
    keep = rows[first[:n].long()]
    return ci[keep], keep
