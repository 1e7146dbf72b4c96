"""Deterministic synthetic SemanticKITTI-shaped / NuScenes-shaped frames.

No dataset ships with this repo (and none can be fetched), so the bench and the
parity tests run on frames generated here.  The generator follows the
reference's own data path as far as it can be followed without real data:

  * LiDAR -> image projection with a KITTI-like ``P2 @ Tr`` in float32, strict
    frustum test, (row, col) order: ``data/semantic_kitti/preprocess.py:32-33,
    86-89,109-116``.
  * ``img_indices = points_img.astype(int64)``:
    ``data/semantic_kitti/semantic_kitti_dataloader.py:193``.
  * voxel coordinates = ``augment_and_scale_3d(points, scale=20,
    full_scale=4096)`` with no augmentation (``data/utils/augmentation_3d.py:
    41-44``), ``astype(int64)``, in-range mask, dedupe keeping the first point
    of each voxel in sorted-key order (``semantic_kitti_dataloader.py:216-238``).
  * 4-channel feats (x, y, z, intensity): ``preprocess.py:121``.
"""
from __future__ import annotations

import numpy as np

# Fractions of labelled points per class on sequence 08
# (reference notebooks/dataset_stats.ipynb cell 22), in the 20-class training
# id order of data/semantic_kitti/semantic_kitti_label.yaml (0 = ignored).
_CLASS_FREQ = np.array([
    0.02,      # 0 unlabeled / ignored
    0.084328,  # car
    0.000577,  # bicycle
    0.000682,  # motorcycle
    0.001334,  # truck
    0.006224,  # other-vehicle
    0.001671,  # person
    0.001700,  # bicyclist
    0.000066,  # motorcyclist
    0.283110,  # road
    0.012861,  # parking
    0.107630,  # sidewalk
    0.000889,  # other-ground
    0.083264,  # building
    0.022334,  # fence
    0.269335,  # vegetation
    0.011646,  # trunk
    0.107318,  # terrain
    0.004190,  # pole
    0.000840,  # traffic-sign
], dtype=np.float64)
_CLASS_FREQ = _CLASS_FREQ / _CLASS_FREQ.sum()

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)

SHAPES = {
    # name: (H, W, beams, elev_lo, elev_hi, az_half_deg, az_step_deg, fx, cx, cy)
    "kitti": dict(H=370, W=1226, beams=64, elev=(-24.8, 2.0), az_half=45.0, az_step=0.09, f=718.856, cx=607.1928, cy=185.2157),
    # throughput-only shape (BASELINE configs[2]: 900x1600 image, ~30k points): the beam fan is the part of the
    # sensor that falls inside the camera frustum, sampled so that ~30k voxels survive the 5 cm dedupe
    "nuscenes": dict(H=900, W=1600, beams=64, elev=(-19.0, 12.0), az_half=35.0, az_step=0.05, f=1266.4, cx=816.27, cy=491.5),
}


def _proj_matrix(shape):
    """KITTI-like P2 @ Tr (velodyne -> camera 2), float32 as preprocess.py:32-33."""
    s = SHAPES[shape]
    P2 = np.array([[s["f"], 0.0, s["cx"], 45.38], [0.0, s["f"], s["cy"], -0.1130887], [0.0, 0.0, 1.0, 0.003779761]])
    Tr = np.identity(4)
    Tr[:3, :4] = np.array([
        [4.2768028e-04, -9.9996725e-01, -8.0844917e-03, -1.1984599e-02],
        [-7.2106265e-03, 8.0811985e-03, -9.9994132e-01, -5.4039847e-02],
        [9.9997386e-01, 4.8594858e-04, -7.2069002e-03, -2.9219686e-01]])
    return (P2 @ Tr).astype(np.float32)


def _raycast_scene(rng, shape):
    s = SHAPES[shape]
    elev = np.deg2rad(np.linspace(s["elev"][0], s["elev"][1], s["beams"]))
    az = np.deg2rad(np.arange(-s["az_half"], s["az_half"], s["az_step"]))
    E, A = np.meshgrid(elev, az, indexing="ij")
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    t = np.full(d.shape[0], np.inf)
    # ground plane z = -1.73 m (sensor height)
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(d[:, 2] < 0, -1.73 / d[:, 2], np.inf)
    t = np.minimum(t, tg)
    # 25 axis-aligned wall patches
    for _ in range(25):
        axis = int(rng.integers(0, 2))          # wall normal along x or y
        c = rng.uniform(4.0, 60.0) if axis == 0 else rng.uniform(-30.0, 30.0)
        lo = rng.uniform(-30.0, 20.0) if axis == 0 else rng.uniform(3.0, 50.0)
        ext = rng.uniform(2.0, 15.0)
        height = rng.uniform(1.0, 6.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            tw = c / d[:, axis]
        p = d * tw[:, None]
        other = 1 - axis
        ok = (tw > 0) & (p[:, other] >= lo) & (p[:, other] <= lo + ext) & (p[:, 2] >= -1.73) & (p[:, 2] <= -1.73 + height)
        t = np.where(ok & (tw < t), tw, t)
    keep = np.isfinite(t) & (t <= 80.0) & (t > 0.5)
    pts = d[keep] * t[keep, None] + rng.normal(0.0, 0.01, size=(int(keep.sum()), 3))
    return pts.astype(np.float32)


def project_points(points, proj_matrix, width, height):
    """LiDAR -> image projection, statement for statement preprocess.py:108-116.

    Returns (keep mask over `points`, points_img (row, col) float for the kept points)."""
    keep = points[:, 0] > 0
    hc = np.concatenate([points[keep], np.ones([int(keep.sum()), 1], dtype=np.float32)], axis=1)
    img_points = (proj_matrix @ hc.T).T
    img_points = img_points[:, :2] / np.expand_dims(img_points[:, 2], axis=1)
    in_img = (img_points[:, 0] > 0) * (img_points[:, 1] > 0) * (img_points[:, 0] < width) * (img_points[:, 1] < height)
    keep[keep] = in_img
    return keep, np.fliplr(img_points)[in_img]


def scale_points_to_voxels(points, scale, full_scale):
    """augment_and_scale_3d without augmentation (augmentation_3d.py:41-44) + int cast and
    in-range mask (semantic_kitti_dataloader.py:216-225).  Returns (coords int64, valid mask)."""
    coords = points * scale
    coords -= coords.min(0)
    coords = coords.astype(np.int64)
    valid = (coords.min(1) >= 0) * (coords.max(1) < full_scale)
    return coords, valid


def make_frame(seed: int, shape: str = "kitti", scale: int = 20, full_scale: int = 4096, max_points=None):
    """One synthetic frame as the dict SemanticKITTISCN.__getitem__ would return.

    Keys: coords (N,3) int64, feats (N,4) f32, seg_label (N,) int64,
    img (3,H,W) f32, img_indices (N,2) int64 (row, col)."""
    s = SHAPES[shape]
    rng = np.random.default_rng(seed)
    points = _raycast_scene(rng, shape)
    intensity = rng.uniform(0.0, 1.0, size=(points.shape[0], 1)).astype(np.float32)
    keep, img_points = project_points(points, _proj_matrix(shape), s["W"], s["H"])
    points = points[keep]
    feats = np.concatenate([points, intensity[keep]], 1).astype(np.float32)
    seg_label = rng.choice(20, size=points.shape[0], p=_CLASS_FREQ).astype(np.int64)
    img_indices = img_points.astype(np.int64)
    coords, valid = scale_points_to_voxels(points, scale, full_scale)
    coords, feats, seg_label, img_indices = coords[valid], feats[valid], seg_label[valid], img_indices[valid]
    # dedupe: first point per voxel in sorted-key order
    key = (coords[:, 0] * full_scale + coords[:, 1]) * full_scale + coords[:, 2]
    _, inds = np.unique(key, return_index=True)
    if max_points is not None and inds.shape[0] > max_points:
        inds = np.sort(rng.choice(inds, size=max_points, replace=False))
    img = rng.uniform(0.0, 1.0, size=(s["H"], s["W"], 3)).astype(np.float32)
    img = (img - IMAGENET_MEAN) / IMAGENET_STD
    return {
        "coords": coords[inds],
        "feats": feats[inds],
        "seg_label": seg_label[inds],
        "img": np.ascontiguousarray(np.moveaxis(img, -1, 0)),
        "img_indices": img_indices[inds],
    }


def make_batch(seeds, shape: str = "kitti", max_points=None):
    """Collated numpy batch following data/collate.py:37-82 (coords get the batch index appended)."""
    frames = [make_frame(s, shape, max_points=max_points) for s in seeds]
    locs = [np.concatenate([f["coords"], np.full((f["coords"].shape[0], 1), i, dtype=np.int64)], 1) for i, f in enumerate(frames)]
    return {
        "coords": np.concatenate(locs, 0),
        "feats": np.concatenate([f["feats"] for f in frames], 0),
        "seg_label": np.concatenate([f["seg_label"] for f in frames], 0),
        "img": np.stack([f["img"] for f in frames]),
        "img_indices": [f["img_indices"] for f in frames],
    }
