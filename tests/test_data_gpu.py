"""Dataset-side voxelisation + batch packing on the device against the reference's numpy statements
(data/semantic_kitti/semantic_kitti_dataloader.py:216-253 with sparse_quantize restated as numpy.unique -- torchsparse is not
importable, PARITY UNPINNED for that one call -- and data/collate.py:37-82), bit for bit; then the batch goes through the model
and through the eval scatter-back, and the result equals the host-built batch's."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _raw_frame(seed, n=6000, out_of_range=0):
    rng = np.random.default_rng(seed)
    pts = (rng.uniform(-1, 1, size=(n, 3)) * np.array([40, 25, 2.5])).astype(np.float32)
    pts[:, 0] = np.abs(pts[:, 0])
    pts[: n // 3] = np.round(pts[: n // 3] * 4) / 4          # clumps: several points per 5 cm voxel
    if out_of_range:
        pts[-out_of_range:, 1] += 400.0                         # > 4096 voxels away from the rest after scaling by 20
    feats = np.concatenate([pts, rng.uniform(0, 1, (n, 1)).astype(np.float32)], 1)
    lab = rng.integers(0, 20, n).astype(np.int64)
    idx = np.stack([rng.integers(0, 370, n), rng.integers(0, 1226, n)], 1).astype(np.int64)
    img = rng.standard_normal((3, 370, 1226)).astype(np.float32)
    return dict(points=pts, feats=feats, seg_label=lab, img_indices=idx, img=img)


def _reference_getitem(f, scale=20, full_scale=4096):
    """Statements of semantic_kitti_dataloader.py:216-253 (no augmentation), sparse_quantize as np.unique."""
    coords = f["points"] * scale
    coords -= coords.min(0)
    coords = coords.astype(np.int64)
    voxel_valid_idxs = (coords.min(1) >= 0) * (coords.max(1) < full_scale)
    voxel_coords = coords[voxel_valid_idxs]
    voxel_feats, voxel_seg_label, voxel_img_indices = f["feats"][voxel_valid_idxs], f["seg_label"][voxel_valid_idxs], f["img_indices"][voxel_valid_idxs]
    key = (voxel_coords[:, 0] * full_scale + voxel_coords[:, 1]) * full_scale + voxel_coords[:, 2]
    _, inds, inverse = np.unique(key, return_index=True, return_inverse=True)
    return dict(voxel_coords=voxel_coords, coords=voxel_coords[inds], feats=voxel_feats[inds], seg_label=voxel_seg_label[inds],
                img_indices=voxel_img_indices[inds], orig_seg_label=f["seg_label"], sparse_orig_points_idx=voxel_valid_idxs[inds],
                inverse_map=inverse, img=f["img"])


def _to_device(f):
    return {k: torch.from_numpy(v).cuda() for k, v in f.items()}


@pytest.mark.parametrize("bad", [0, 37])
def test_voxelize_frames_matches_the_reference_statements(bad):
    from fusiontransformer_amd.data.voxelize import voxelize_frames
    raw = [_raw_frame(1, 6000, bad), _raw_frame(2, 4500, 0), _raw_frame(3, 1, 0)]
    ref = [_reference_getitem({k: v.copy() for k, v in f.items()}) for f in raw]
    out = voxelize_frames([_to_device(f) for f in raw])
    assert len(out) == 3
    for r, o in zip(ref, out):
        for k in ("voxel_coords", "coords", "feats", "seg_label", "img_indices", "inverse_map", "sparse_orig_points_idx", "orig_seg_label"):
            assert np.array_equal(o[k].cpu().numpy(), r[k]), k
        assert o["coords"].shape[0] < o["voxel_coords"].shape[0] or o["voxel_coords"].shape[0] == 1   # the clumps really dedupe
        # inverse_map maps every (valid) point to its voxel: coords[inverse] == voxel_coords
        assert torch.equal(o["coords"][o["inverse_map"]], o["voxel_coords"])


def test_device_collate_equals_host_collate_and_feeds_model_and_eval():
    from fusiontransformer_amd.data.collate import collate_scn_base
    from fusiontransformer_amd.data.voxelize import collate_device, voxelize_frames
    from fusiontransformer_amd.evaluate import Evaluator, validate_batch
    from fusiontransformer_amd.models.build import build_model
    from tests.helpers import small_cfg
    raw = [_raw_frame(5, 5000), _raw_frame(6, 3000)]
    ref = [_reference_getitem({k: v.copy() for k, v in f.items()}) for f in raw]
    for r in ref:
        r["seq"], r["filename"] = "08", "000000"
    host = collate_scn_base(ref, output_orig=True)
    frames = voxelize_frames([dict(_to_device(f), seq="08", filename="000000") for f in raw])
    dev = collate_device(frames, output_orig=True)
    assert torch.equal(dev["lidar"].C.cpu(), host["lidar"].C) and torch.equal(dev["lidar"].F.cpu(), host["lidar"].F)
    assert torch.equal(dev["seg_label"].cpu(), host["seg_label"]) and torch.equal(dev["img"].cpu(), host["img"])
    idx, frame = dev["img_indices"]
    assert np.array_equal(idx.cpu().numpy(), np.concatenate(host["img_indices"], 0))
    assert np.array_equal(frame.cpu().numpy(), np.concatenate([np.full(len(a), i) for i, a in enumerate(host["img_indices"])]))
    for a, b in zip(dev["inverse_map"], host["inverse_map"]):
        assert np.array_equal(a.cpu().numpy(), b)
    assert dev["seq"] == host["seq"] and dev["filename"] == host["filename"]
    # the two batches give the same predictions and the same evaluation
    cfg = small_cfg("middle")
    torch.manual_seed(0)
    model, _, _ = build_model(cfg)
    model = model.cuda().eval()
    labels = np.arange(20)
    with torch.no_grad():
        out_d = model({"img": dev["img"], "img_indices": dev["img_indices"], "lidar": dev["lidar"]})
        hb = {"img": host["img"].cuda(), "img_indices": host["img_indices"], "lidar": host["lidar"].cuda()}
        out_h = model(hb)
    for k in out_h:
        assert torch.equal(out_d[k], out_h[k]), k
    names = ["c%d" % i for i in range(20)]
    ev_d, ev_h = [Evaluator(names, labels=labels) for _ in range(3)], [Evaluator(names, labels=labels) for _ in range(3)]
    rd = validate_batch(out_d, dev, labels, *ev_d, want_preds=True)
    rh = validate_batch(out_h, host, labels, *ev_h, want_preds=True)
    assert int(rd["bad_index_flag"].item()) == 0
    for k in ("pred_3d", "pred_2d", "pred_ensemble"):
        assert torch.equal(rd[k], rh[k]), k
    for a, b in zip(ev_d, ev_h):
        assert torch.equal(a.mat, b.mat) and int(a.mat.sum().item()) > 0
    # and against numpy: prediction of an original point = prediction of its voxel (validate.py:10-11 map_sparse_to_org)
    p3 = out_d["lidar_seg_logit"].argmax(1).cpu().numpy()
    off = 0
    exp = []
    for r in ref:
        exp.append(p3[off:off + r["coords"].shape[0]][r["inverse_map"]])
        off += r["coords"].shape[0]
    assert np.array_equal(rd["pred_3d"].cpu().numpy(), np.concatenate(exp))


def test_voxelize_frames_with_the_augmentation_branch():
    """The 3-D augmentation on the device inside voxelize_frames (rotation by libftx, translation in float64 like numpy) against the
    dataloader's statements with the oracle's restatement of augment_and_scale_3d (itself pinned by the reference function's output,
    tests/test_golden.py): every key bit for bit, draws made by draw_augmentation_3d from a seeded numpy.random."""
    from fusiontransformer_amd.data.augment import draw_augmentation_3d
    from fusiontransformer_amd.data.voxelize import voxelize_frames
    from oracle import ft_oracle as O
    raw = [_raw_frame(7, 6000, 11), _raw_frame(8, 3000, 0)]
    np.random.seed(5)
    draws = [draw_augmentation_3d(noisy_rot=0.1, flip_x=0.5, flip_y=0.5, rot_z=2 * np.pi, transl=True) for _ in raw]
    ref = []
    for f, (rot, u) in zip(raw, draws):
        g = {k: v.copy() for k, v in f.items()}
        coords = O.augment_and_scale_3d_np(g["points"], 20, 4096, rot, u).astype(np.int64)
        valid = (coords.min(1) >= 0) * (coords.max(1) < 4096)
        vc = coords[valid]
        key = (vc[:, 0] * 4096 + vc[:, 1]) * 4096 + vc[:, 2]
        _, inds, inverse = np.unique(key, return_index=True, return_inverse=True)
        ref.append(dict(voxel_coords=vc, coords=vc[inds], feats=g["feats"][valid][inds], seg_label=g["seg_label"][valid][inds],
                        img_indices=g["img_indices"][valid][inds], inverse_map=inverse))
    out = voxelize_frames([_to_device(f) for f in raw], augment=draws)
    for r, o in zip(ref, out):
        for k in r:
            assert np.array_equal(o[k].cpu().numpy(), r[k]), k
    plain = voxelize_frames([_to_device(f) for f in raw])
    assert not torch.equal(plain[0]["voxel_coords"], out[0]["voxel_coords"])          # the augmentation really moved the voxels


def test_image_side_augmentation_matches_the_dataloader_statements():
    """Bottom crop + point filter + shift, int cast, left-right flip with the column update, normalisation, HWC -> CHW
    (semantic_kitti_dataloader.py:166-212, restated in numpy here: that class is not importable -- PARITY UNPINNED)."""
    from fusiontransformer_amd.data.augment import augment_image, draw_augmentation_2d
    rng = np.random.default_rng(3)
    H, W, n = 370, 1226, 5000
    image = rng.random((H, W, 3)).astype(np.float32)
    points_img = np.stack([rng.uniform(0, H, n), rng.uniform(0, W, n)], 1).astype(np.float32)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    for seed, crop, fliplr in ((1, (480, 302), 0.5), (2, (480, 302), 1.0), (3, None, 1.0), (4, None, None)):
        np.random.seed(seed)
        box, flip = draw_augmentation_2d((W, H), crop, fliplr)
        # the dataloader's statements, with the same seed
        np.random.seed(seed)
        img, pi, keep = image, points_img.copy(), np.ones(n, dtype=bool)
        if crop is not None:
            left = int(np.random.rand() * (W + 1 - crop[0])); right = left + crop[0]; top = H - crop[1]; bottom = H
            keep = (pi[:, 0] >= top) & (pi[:, 0] < bottom) & (pi[:, 1] >= left) & (pi[:, 1] < right)
            img = img[top:bottom, left:right]
            pi = pi[keep]
            pi[:, 0] -= top
            pi[:, 1] -= left
        idx = pi.astype(np.int64)
        if (fliplr is not None) and (np.random.rand() < fliplr):
            img = np.ascontiguousarray(np.fliplr(img))
            idx[:, 1] = img.shape[1] - 1 - idx[:, 1]
        img = (img - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)
        want = np.moveaxis(img, -1, 0)
        got_img, got_idx, got_keep = augment_image(torch.from_numpy(image).cuda(), torch.from_numpy(points_img).cuda(), box, flip, (mean, std))
        assert np.array_equal(got_keep.cpu().numpy(), keep) and np.array_equal(got_idx.cpu().numpy(), idx), seed
        assert got_img.shape == want.shape and np.array_equal(got_img.cpu().numpy(), want), seed
