"""The dataset's augmentation branch on the device (the part that is off by default in this fork:
config/FusionTransformerConfig.py:88-97).

3-D (data/utils/augmentation_3d.py:4-53, called from semantic_kitti_dataloader.py:216-217): noisy rotation, axis flips, rotation
about z, random translation inside the receptive field.  The RANDOM DRAWS stay on the host -- `draw_augmentation_3d` consumes
`numpy.random` in exactly the order the reference does, so a seeded run picks the same matrix and offset --; what is applied to the
points runs on the GPU with numpy's float32 / float64 rounding (`augment_and_scale_3d`): bit-exact against the reference function run in
the build container (tests/golden/voxel_coords_augmented.npz).

2-D (semantic_kitti_dataloader.py:166-212): bottom crop with the point filter, left-right flip with the column update, normalisation,
HWC -> CHW (`augment_image`); `draw_augmentation_2d` makes the two draws.  Colour jitter is torchvision's (not installed, not used by any
live config) and is not provided.  That dataloader class cannot be imported here (torchvision, torchsparse), so the 2-D part is
checked against a numpy restatement of its statements: PARITY UNPINNED, though every operation is an index or a float32 subtraction
and division."""
from __future__ import annotations

import numpy as np
import torch

from .. import functional as spf

__all__ = ["draw_augmentation_3d", "augment_and_scale_3d", "draw_augmentation_2d", "augment_image"]


def draw_augmentation_3d(noisy_rot=0.0, flip_x=0.0, flip_y=0.0, rot_z=0.0, transl=False, rng=np.random):
    """The random numbers of one call of the reference's augment_and_scale_3d, drawn in its order (augmentation_3d.py:22-51):
    randn(3, 3) for the noisy rotation, one randint(0, 2) per enabled flip, rand() for the z angle, rand(3) for the translation.
    Returns (rot (3,3) float32 or None, transl_u (3,) float64 or None).  `rng`: numpy.random or a RandomState."""
    rot = None
    if noisy_rot > 0 or flip_x > 0 or flip_y > 0 or rot_z > 0:
        rot = np.eye(3, dtype=np.float32)
        if noisy_rot > 0:
            rot += rng.randn(3, 3) * noisy_rot                 # float64 noise added into the float32 matrix
        if flip_x > 0:
            rot[0][0] *= rng.randint(0, 2) * 2 - 1
        if flip_y > 0:
            rot[1][1] *= rng.randint(0, 2) * 2 - 1
        if rot_z > 0:
            theta = rng.rand() * rot_z
            zr = np.array([[np.cos(theta), -np.sin(theta), 0], [np.sin(theta), np.cos(theta), 0], [0, 0, 1]], dtype=np.float32)
            rot = rot.dot(zr)
    u = rng.rand(3) if transl else None
    return rot, u


def augment_and_scale_3d(points: torch.Tensor, scale, full_scale, rot=None, transl_u=None) -> torch.Tensor:
    """points (N,3) float32 on the GPU -> float voxel coordinates, as the reference's function with the draws made by
    draw_augmentation_3d: points . rot (libftx, the sgemm's fused rounding), * scale, translation to the positive octant, and the
    random offset `clip(full_scale - max - 0.001, 0) * u` added in float64 and rounded back to float32 (numpy's in-place add of a
    float64 array into a float32 one)."""
    if not points.is_cuda or points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("augment_and_scale_3d: expected a (N,3) float32 CUDA tensor")
    if rot is not None:
        points = spf.rotate_points(points.contiguous(), rot)
    coords = points * float(scale)
    coords = coords - coords.min(0).values
    if transl_u is not None:
        room = torch.clamp((float(full_scale) - coords.max(0).values) - 0.001, min=0.0)          # float32 throughout, like numpy's weak scalars
        offset = room.double() * torch.as_tensor(np.asarray(transl_u, dtype=np.float64), device=points.device)
        coords = (coords.double() + offset).float()
    return coords


def draw_augmentation_2d(image_size, bottom_crop=None, fliplr=None, rng=np.random):
    """The draws of semantic_kitti_dataloader.py:170-197 in order: the crop's left edge (if bottom_crop = (width, height)), then the
    flip decision (if fliplr is a probability).  image_size = (width, height) of the uncropped image.  Returns (crop box
    (left, top, right, bottom) or None, flip: bool)."""
    box = None
    if bottom_crop is not None:
        left = int(rng.rand() * (image_size[0] + 1 - bottom_crop[0]))
        box = (left, image_size[1] - bottom_crop[1], left + bottom_crop[0], image_size[1])
    flip = (fliplr is not None) and bool(rng.rand() < fliplr)
    return box, flip


def augment_image(image: torch.Tensor, points_img: torch.Tensor, box=None, flip=False, normalizer=None):
    """image (H,W,3) float32 in [0,1] on the GPU, points_img (N,2) float (row, col).  Returns (img (3,h,w) float32, img_indices (K,2)
    int64, keep (N,) bool): crop + point filter + shift (:176-190), truncation to int64 (:193), flip with `w - 1 - col` (:201-203),
    (image - mean) / std (:206-210), HWC -> CHW (:212).  `keep` is what the caller applies to points / feats / labels (:187-190)."""
    keep = torch.ones((points_img.shape[0],), dtype=torch.bool, device=points_img.device)
    pi = points_img
    if box is not None:
        left, top, right, bottom = box
        keep = (pi[:, 0] >= top) & (pi[:, 0] < bottom) & (pi[:, 1] >= left) & (pi[:, 1] < right)
        image = image[top:bottom, left:right]
        pi = pi[keep].clone()
        pi[:, 0] -= top
        pi[:, 1] -= left
    idx = pi.to(torch.int64)
    if flip:
        image = torch.flip(image, dims=(1,))
        idx = idx.clone()
        idx[:, 1] = image.shape[1] - 1 - idx[:, 1]
    if normalizer is not None:
        mean, std = normalizer
        mean = torch.as_tensor(np.asarray(mean, dtype=np.float32), device=image.device)
        std = torch.as_tensor(np.asarray(std, dtype=np.float32), device=image.device)
        image = (image - mean) / std
    return image.permute(2, 0, 1).contiguous(), idx, keep
