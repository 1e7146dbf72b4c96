"""Dataset-side voxelisation, dedupe and batch packing on the GPU.

The host work of SemanticKITTISCN.__getitem__ (data/semantic_kitti/semantic_kitti_dataloader.py:216-253) --
`augment_and_scale_3d` without augmentation (data/utils/augmentation_3d.py:41-44), the int64 cast, the in-range mask,
torchsparse `sparse_quantize(..., return_index=True, return_invs=True)` and the per-key gathers -- and of
`collate_scn_base` (data/collate.py:37-82), on device tensors:

  * `points_to_voxels`      one frame, coordinates only (kept from round 1; bit-exact with tests/golden/voxel_coords.npz);
  * `voxelize_frames`       every key `__getitem__` produces, for a list of frames, with ONE host read for the whole batch
                            (the per-frame voxel counts); everything before it is queued without synchronising;
  * `collate_device`        the batch the model consumes: SparseTensor with the batch column, labels, stacked images and the
                            packed `(idx, frame)` pair `Net2DBillinear` takes directly, plus the `output_orig` keys
                            (`inverse_map`, `orig_seg_label`, `sparse_orig_points_idx`) that `evaluate.validate_batch` consumes.

`sparse_quantize` is torchsparse v1.1.0 (not importable here, PARITY UNPINNED): restated as numpy.unique over the ravelled
voxel key -- unique voxels in ascending key order, `inds` = first point of each voxel in the input order (stable sort),
`inverse` = rank of every point's voxel (np.unique's return_index / return_inverse).  tests/test_data_gpu.py checks the device
path against that numpy statement bit for bit."""
from __future__ import annotations

import numpy as np
import torch

from .. import functional as spf
from ..sparse import HostRead, SparseTensor

_SENTINEL = (1 << 60) - 1      # sorts after every real key (keys < full_scale^3 <= 2^36); libftx sorts bits [0, 60)


def _scale_cast_mask(points: torch.Tensor, scale: int, full_scale: int, aug=None):
    if aug is not None:                                  # (rot, transl_u) from data.augment.draw_augmentation_3d
        from .augment import augment_and_scale_3d
        coords = augment_and_scale_3d(points, scale, full_scale, aug[0], aug[1])
    else:
        coords = points * float(scale)                   # float32 multiply, as numpy does
        coords = coords - coords.min(0).values           # translate to the positive octant (augmentation_3d.py:41-44)
    ci = coords.to(torch.int64)                          # astype(np.int64): truncation
    valid = (ci.min(1).values >= 0) & (ci.max(1).values < full_scale)
    key = (ci[:, 0] * full_scale + ci[:, 1]) * full_scale + ci[:, 2]
    return ci, valid, key


def points_to_voxels(points: torch.Tensor, scale: int = 20, full_scale: int = 4096):
    """points (N,3) float32 on the GPU -> (coords (M,3) int64 of the kept, deduped voxels,
    keep_index (M,) int64 rows of `points` they come from, in sorted-key order like np.unique)."""
    if not points.is_cuda or points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("points_to_voxels: expected a (N,3) float32 CUDA tensor")
    ci, valid, key = _scale_cast_mask(points, scale, full_scale)
    rows = torch.nonzero(valid).squeeze(1)               # single frame, dataloader side: a host sync here is fine
    uniq, first, cnt = spf.unique_sorted(key[rows].contiguous())   # ascending key, first occurrence of each
    n = int(cnt.item())
    keep = rows[first[:n].long()]
    return ci[keep], keep


def voxelize_frames(frames, scale: int = 20, full_scale: int = 4096, augment=None):
    """The voxelisation half of SemanticKITTISCN.__getitem__ for a list of frames, on the device.

    `augment`: optional list with one `(rot, transl_u)` pair per frame (data.augment.draw_augmentation_3d: the reference's 3-D
    augmentation with its random draws made on the host in the reference's order); None = no augmentation, this fork's default.

    frames: list of dicts with device tensors `points` (N,3) f32, `feats` (N,C) f32, `seg_label` (N,) int64,
    `img_indices` (N,2) int64 (row, col) and anything else (`img`, `seq`, `filename`), which is passed through.
    Returns a list of dicts with the reference's keys: `voxel_coords`, `coords`, `feats`, `seg_label`, `img_indices`,
    `orig_seg_label`, `sparse_orig_points_idx`, `inverse_map` (device tensors).

    All kernels of all frames are queued first; the per-frame voxel counts (and out-of-range point counts) then come back in ONE
    pinned-memory read."""
    queued = []
    if augment is not None and len(augment) != len(frames):
        raise ValueError("voxelize_frames: `augment` needs one (rot, transl_u) pair per frame")
    for fi, f in enumerate(frames):
        pts = f["points"]
        if not pts.is_cuda or pts.dtype != torch.float32 or pts.dim() != 2 or pts.shape[1] != 3:
            raise ValueError("voxelize_frames: `points` must be a (N,3) float32 CUDA tensor")
        n = pts.shape[0]
        for k in ("feats", "seg_label", "img_indices"):
            if f[k].shape[0] != n or not f[k].is_cuda:
                raise ValueError("voxelize_frames: `%s` must be a CUDA tensor with one row per point" % k)
        ci, valid, key = _scale_cast_mask(pts, scale, full_scale, None if augment is None else augment[fi])
        key = torch.where(valid, key, torch.full_like(key, _SENTINEL)).contiguous()   # out-of-range points sort last, as one extra group
        uniq, first, cnt = spf.unique_sorted(key)
        inverse = spf.sorted_rank(uniq, cnt, key)
        n_bad = (~valid).sum().to(torch.int32).view(1)
        queued.append((f, ci, valid, first, inverse, cnt, n_bad))
    if not queued:
        return []
    sizes = HostRead(torch.cat([torch.cat([q[5], q[6]]) for q in queued])).values()      # the one host read of the batch
    out = []
    for i, (f, ci, valid, first, inverse, _, _) in enumerate(queued):
        n_groups, n_bad = sizes[2 * i], sizes[2 * i + 1]
        n_vox = n_groups - (1 if n_bad > 0 else 0)
        inds = first[:n_vox].long()                      # rows of the frame, ascending key order
        if n_bad == 0:                                   # the reference's voxel_valid_idxs is all True (it asserts so in validation)
            voxel_coords, inv, inds_valid, lab_valid = ci, inverse.long(), inds, f["seg_label"]
        else:                                            # compact to the valid points, as `coords[voxel_valid_idxs]` does
            pos = torch.cumsum(valid.to(torch.int64), 0) - 1
            voxel_coords, inv, inds_valid, lab_valid = ci[valid], inverse.long()[valid], pos[inds], f["seg_label"][valid]
        d = {k: v for k, v in f.items() if k not in ("points", "feats", "seg_label", "img_indices")}
        d.update({
            "voxel_coords": voxel_coords,
            "coords": ci[inds],
            "feats": f["feats"][inds],
            "seg_label": f["seg_label"][inds],
            "img_indices": f["img_indices"][inds],
            "orig_seg_label": f["seg_label"],
            "orig_seg_label_valid": lab_valid,           # labels of the points `inverse_map` covers (== orig_seg_label when all are in range)
            "sparse_orig_points_idx": torch.ones((n_vox,), dtype=torch.bool, device=ci.device),   # voxel_valid_idxs[sparse_unique_inds]
            "inverse_map": inv,
            "unique_inds": inds_valid,                   # sparse_unique_inds: rows of voxel_coords
        })
        out.append(d)
    return out


def collate_device(frame_dicts, output_orig: bool = False, output_image: bool = True):
    """collate_scn_base (data/collate.py:6-86) on device tensors: no host copy, no per-frame H2D of the image indices.

    `lidar` is a SparseTensor with int32 [x, y, z, batch] coordinates, `img_indices` the packed (idx (sum N, 2) int64,
    frame (sum N,) int32) pair that Net2DBillinear consumes as is.  With output_orig the per-frame `inverse_map` /
    `orig_seg_label` / `sparse_orig_points_idx` lists of the reference are returned, and `inverse_map_packed` /
    `orig_seg_label_packed` as well: the same maps with the frame offsets added and concatenated, the form
    `evaluate.validate_batch` and ftx_eval_scatter_back take."""
    if not frame_dicts:
        raise ValueError("collate_device: empty batch")
    locs, feats, labels, frame_ids = [], [], [], []
    for i, d in enumerate(frame_dicts):
        c = d["coords"]
        locs.append(torch.cat([c.to(torch.int32), torch.full((c.shape[0], 1), i, dtype=torch.int32, device=c.device)], 1))
        feats.append(d["feats"])
        if "seg_label" in d:
            labels.append(d["seg_label"])
        frame_ids.append(torch.full((c.shape[0],), i, dtype=torch.int32, device=c.device))
    out = {"lidar": SparseTensor(coords=torch.cat(locs, 0).contiguous(), feats=torch.cat(feats, 0).contiguous())}
    if labels:
        out["seg_label"] = torch.cat(labels, 0)
    if output_image:
        out["img"] = torch.stack([d["img"] for d in frame_dicts])
        out["img_indices"] = (torch.cat([d["img_indices"] for d in frame_dicts], 0).contiguous(), torch.cat(frame_ids, 0).contiguous())
    if output_orig:
        out["orig_seg_label"] = [d["orig_seg_label"] for d in frame_dicts]
        out["sparse_orig_points_idx"] = [d["sparse_orig_points_idx"] for d in frame_dicts]
        out["inverse_map"] = [d["inverse_map"] for d in frame_dicts]
        offs = np.cumsum([0] + [int(d["coords"].shape[0]) for d in frame_dicts[:-1]])
        out["inverse_map_packed"] = torch.cat([d["inverse_map"] + int(o) for d, o in zip(frame_dicts, offs)], 0).contiguous()
        out["orig_seg_label_packed"] = torch.cat([d["orig_seg_label_valid"] for d in frame_dicts], 0).contiguous()
    for k in ("seq", "filename"):
        if all(k in d for d in frame_dicts):
            out[k] = [d[k] for d in frame_dicts]
    out["voxel_coords"] = [d["voxel_coords"] for d in frame_dicts]
    return out
