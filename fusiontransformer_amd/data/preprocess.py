"""Offline preprocessing on the GPU and the on-disk scan format.

Reference: FusionTransformer/data/semantic_kitti/preprocess.py -- `DummyDataset.__getitem__` (lines 93-126)
projects every LiDAR point into the camera image and keeps the ones inside it, `preprocess()` (lines 130-170)
writes one pickle per scan: {'points' (N,3) f32, 'feats' (N,4) f32, 'seg_labels' (N,) int16,
'points_img' (N,2) f32 (row, col), 'lidar_path', 'camera_path', 'image_size'}.  The dataloader reads those files
back (semantic_kitti_dataloader.py:150-153).

`project_scan` does the projection with `ftx_project_points`; `write_scan` / `read_scan` are the file contract.
`read_scan` never executes anything from the file: it unpickles with a whitelist that admits only what the
reference writer produces (numpy arrays / dtypes / scalars and builtin containers)."""
from __future__ import annotations

import io
import pickle

import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, req

SCAN_KEYS = ("points", "feats", "seg_labels", "points_img", "lidar_path", "camera_path", "image_size")


def project_points(points: torch.Tensor, proj_matrix: torch.Tensor, width: int, height: int):
    """(keep (N,) bool, rowcol (N,2) float32) for LiDAR points (N,3) on the GPU: preprocess.py:108-116."""
    L = _lib.load()
    req(points, torch.float32, "project_points points", 2)
    req(proj_matrix, torch.float32, "project_points proj_matrix", 2)
    if points.shape[1] != 3 or tuple(proj_matrix.shape) != (3, 4):
        raise ValueError("project_points: points must be (N,3) and proj_matrix (3,4)")
    n = points.shape[0]
    keep = torch.empty((n,), dtype=torch.uint8, device=points.device)
    rowcol = torch.empty((n, 2), dtype=torch.float32, device=points.device)
    check(L.ftx_project_points(ptr(points), n, ptr(proj_matrix), int(width), int(height), ptr(keep), ptr(rowcol),
                               torch.cuda.current_stream(points.device).cuda_stream), "ftx_project_points")
    return keep.bool(), rowcol


def project_scan(scan: np.ndarray, label: np.ndarray, proj_matrix: np.ndarray, image_size, device="cuda"):
    """The dict DummyDataset.__getitem__ builds from one raw scan (N,4) f32 + labels (N,) uint32 (preprocess.py:96-124)."""
    scan_d = torch.from_numpy(np.ascontiguousarray(scan, dtype=np.float32)).to(device)
    keep, rowcol = project_points(scan_d[:, :3].contiguous(), torch.from_numpy(np.ascontiguousarray(proj_matrix, dtype=np.float32)).to(device),
                                  int(image_size[0]), int(image_size[1]))
    idx = torch.nonzero(keep).squeeze(1)     # offline tool: the one host sync sizes the outputs
    label_d = torch.from_numpy((np.asarray(label).reshape(-1) & 0xFFFF).astype(np.int32)).to(device)
    return {
        "seg_label": label_d[idx].to(torch.int16).cpu().numpy(),
        "points": scan_d[idx, :3].cpu().numpy(),
        "feats": scan_d[idx].cpu().numpy(),
        "points_img": rowcol[idx].cpu().numpy(),
        "image_size": np.array(image_size),
    }


def write_scan(path, scan_data: dict):
    """preprocess.py:150-163: one pickle per scan, exactly the reference's keys and array types."""
    missing = [k for k in SCAN_KEYS if k not in scan_data]
    if missing:
        raise ValueError("write_scan: missing keys %s" % missing)
    out = {
        "points": np.asarray(scan_data["points"], dtype=np.float32), "feats": np.asarray(scan_data["feats"], dtype=np.float32),
        "seg_labels": np.asarray(scan_data["seg_labels"], dtype=np.int16), "points_img": np.asarray(scan_data["points_img"], dtype=np.float32),
        "lidar_path": str(scan_data["lidar_path"]), "camera_path": str(scan_data["camera_path"]),
        "image_size": tuple(int(v) for v in scan_data["image_size"]),
    }
    with open(path, "wb") as f:
        pickle.dump(out, f)


_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
}


class _ScanUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("scan file references %s.%s, which a preprocessed scan never contains" % (module, name))


def read_scan(path) -> dict:
    """semantic_kitti_dataloader.py:150-153 without trusting the file: only arrays, scalars, str, tuple, dict load."""
    with open(path, "rb") as f:
        data = _ScanUnpickler(io.BytesIO(f.read())).load()
    if not isinstance(data, dict):
        raise ValueError("read_scan: %s does not hold a scan dict" % path)
    missing = [k for k in SCAN_KEYS if k not in data]
    if missing:
        raise ValueError("read_scan: %s lacks %s" % (path, missing))
    n = data["points"].shape[0]
    if data["points"].shape != (n, 3) or data["feats"].shape != (n, 4) or data["seg_labels"].shape != (n,) or data["points_img"].shape != (n, 2):
        raise ValueError("read_scan: %s has inconsistent array shapes" % path)
    return data
