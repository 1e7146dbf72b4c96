"""The oracle (and the host-side pieces that run on the CPU) against the golden vectors that
tests/golden/make_golden.py produced by running the reference's own importable code."""
import os

import numpy as np
import pytest
import torch

from oracle import ft_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_nearest_index_rule_matches_nn_upsample():
    for key, ref in load("upsample_index.npz").items():
        n_in, n_out = map(int, key.split("_"))
        assert np.array_equal(O.nearest_src_index(n_out, n_in), ref), key


def _load_bilinear(prefix, g, inc, outc, size):
    m = O.BilinearModule(inc, outc, size)
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    m.load_state_dict(sd)
    return m.train()


def test_bilinear_module_down_matches_reference():
    g = load("bilinear_lift.npz")
    down = _load_bilinear("down.", g, 3, 3, (384, 384))
    img = torch.from_numpy(g["img_q"].astype(np.float32) / 256.0)
    out = down(img).detach().numpy()
    np.testing.assert_allclose(out, g["down_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(down.stem[2].running_mean.numpy(), g["down_running_mean_after"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(down.stem[2].running_var.numpy(), g["down_running_var_after"], rtol=1e-6, atol=1e-7)


def test_lift_gather_matches_reference_get_img_feats():
    g = load("bilinear_lift.npz")
    net = O.Net2DBillinear.__new__(O.Net2DBillinear)
    torch.nn.Module.__init__(net)
    net.up = torch.nn.ModuleDict({"5": _load_bilinear("up.", g, 768, 96, (370, 1226))})
    tokens = torch.from_numpy(g["tok_q"].astype(np.float32) / 32.0)
    feats = net.get_img_feats([g["idx0"], g["idx1"]], "5", {"5": tokens}).detach().numpy()
    np.testing.assert_allclose(feats, g["feats"], rtol=1e-5, atol=1e-6)


def test_losses_and_seg_iou_match_reference():
    g = load("losses_metric.npz")
    preds = {k: torch.from_numpy(g[k]) for k in ("lidar_seg_logit", "img_seg_logit", "lidar_seg_logit2", "img_seg_logit2")}
    label = torch.from_numpy(g["label"])
    cw = torch.from_numpy(g["class_weights"])
    l2, l3 = O.fusion_losses(preds, label, cw, float(g["lambda_xm"]), True)
    assert abs(l2.item() - float(g["loss_2d"])) < 1e-6 and abs(l3.item() - float(g["loss_3d"])) < 1e-6
    assert np.array_equal(O.confusion_matrix(preds["lidar_seg_logit"], label, 20).numpy(), g["mat3d"])
    assert np.array_equal(O.confusion_matrix(preds["img_seg_logit"], label, 20).numpy(), g["mat2d"])
    np.testing.assert_allclose(O.iou_from_matrix(torch.from_numpy(g["mat3d"])).numpy(), g["iou3d"], rtol=1e-6, equal_nan=True)
    # the product's host-side loss / metric code (pure torch, device-agnostic) against the same vectors
    from fusiontransformer_amd.models.metric import SegIoU
    from fusiontransformer_amd.trainer import fusion_losses
    p2, p3 = fusion_losses(preds, label, cw, float(g["lambda_xm"]), True)
    assert abs(p2.item() - float(g["loss_2d"])) < 1e-6 and abs(p3.item() - float(g["loss_3d"])) < 1e-6
    m3, m2 = SegIoU(20, name="seg_iou_3d"), SegIoU(20, name="seg_iou_2d")
    m3.update_dict(preds, {"seg_label": label}); m2.update_dict(preds, {"seg_label": label})
    assert np.array_equal(m3.mat.numpy(), g["mat3d"]) and np.array_equal(m2.mat.numpy(), g["mat2d"])
    np.testing.assert_allclose(m2.iou.numpy(), g["iou2d"], rtol=1e-6, equal_nan=True)


def test_torchpack_loss_mix_matches_reference_statements():
    """(1-lambda)*CE + lambda*KL and the ones-with-w[0]=0 default class weights of the DDP trainer
    (modules/SemanticTorchpackTrainer.py:28-32,70-106): oracle and the host-side mirror against the reference's statements."""
    from oracle import ft_oracle as O
    from fusiontransformer_amd.trainer import default_class_weights, fusion_losses
    g, t = load("losses_metric.npz"), load("losses_torchpack.npz")
    preds = {k: torch.from_numpy(g[k]) for k in ("lidar_seg_logit", "img_seg_logit", "lidar_seg_logit2", "img_seg_logit2")}
    label = torch.from_numpy(g["label"])
    for tag in ("cfgw", "defw", "single", "nolam"):
        w, lam, dual = torch.from_numpy(t[tag + "_weights"]), float(t[tag + "_lambda"]), bool(t[tag + "_dual"])
        if tag in ("defw", "nolam"):
            assert torch.equal(w, default_class_weights(20))
        for fn in (O.fusion_losses, fusion_losses):
            l2, l3 = fn(preds, label, w, lam, dual, mix="torchpack")
            assert abs(l2.item() - float(t[tag + "_loss_2d"])) < 1e-6 and abs(l3.item() - float(t[tag + "_loss_3d"])) < 1e-6, (tag, fn)


def test_voxel_coordinates_match_reference_augment_and_scale():
    from fusiontransformer_amd.data.synth import scale_points_to_voxels
    g = load("voxel_coords.npz")
    coords, valid = scale_points_to_voxels(g["points"].copy(), 20, 4096)
    assert np.array_equal(coords, g["coords_int"]) and np.array_equal(valid, g["valid"])


def test_projection_matches_reference_preprocess():
    from fusiontransformer_amd.data.synth import project_points
    g = load("projection.npz")
    keep, pts_img = project_points(g["points"].copy(), g["proj_matrix"], 1226, 370)
    assert np.array_equal(keep, g["keep_idx"])
    assert np.array_equal(pts_img, g["points_img"])
    assert np.array_equal(pts_img.astype(np.int64), g["img_indices"])

def _split(g):
    nv, no = g["n_vox"], g["n_org"]
    cuts = np.cumsum(no)[:-1]
    return nv, np.split(g["inverse_map"], cuts), np.split(g["orig_seg_label"], cuts)


def test_eval_scatter_back_oracle_matches_reference_evaluator():
    g = load("eval_scatter_back.npz")
    nv, inv, gts = _split(g)
    (p3, p2, pe), (m3, m2, me) = O.validate_batch(g["lidar_seg_logit"], g["img_seg_logit"], inv, gts, nv, g["class_labels"])
    assert np.array_equal(p3, g["pred_3d"]) and np.array_equal(p2, g["pred_2d"]) and np.array_equal(pe, g["pred_ensemble"])
    assert np.array_equal(m3, g["conf_3d"]) and np.array_equal(m2, g["conf_2d"]) and np.array_equal(me, g["conf_ensemble"])
    # the reference quirk this pins: unlabeled points (id 0) are re-labelled num_classes = 20, which IS a label id
    # ("other-vehicle", learning id 5), so they are counted as ground truth of that class
    assert g["conf_3d"][5].sum() > (g["orig_seg_label"] == 5).sum()


@pytest.mark.gpu
def test_product_lift_and_resample_match_reference_goldens():
    """The libftx lift gather / nearest resample against the vectors produced by the reference code."""
    from fusiontransformer_amd.models.image_models_billinear import BilinearModule
    g = load("bilinear_lift.npz")
    up = BilinearModule(768, 96, (370, 1226))
    up.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("up.")})
    up = up.cuda().train()
    tokens = torch.from_numpy(g["tok_q"].astype(np.float32) / 32.0).cuda()
    from fusiontransformer_amd import functional as spf
    from fusiontransformer_amd.models.image_models_billinear import pack_img_indices
    grid = up.forward_tokens(tokens, (24, 24))
    idx, frame = pack_img_indices([g["idx0"], g["idx1"]], "cuda")
    feats = spf.lift_gather(grid, idx, frame, 370, 1226).detach().cpu().numpy()
    np.testing.assert_allclose(feats, g["feats"], rtol=1e-4, atol=1e-5)
    down = BilinearModule(3, 3, (384, 384))
    down.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("down.")})
    down = down.cuda().train()
    out = down(torch.from_numpy(g["img_q"].astype(np.float32) / 256.0).cuda()).detach().cpu().numpy()
    np.testing.assert_allclose(out, g["down_out"], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_fused_loss_kernel_matches_reference_goldens():
    """ftx_fusion_loss against the loss values / confusion matrices the reference code produced, and its
    gradients against autograd through the reference's own statements."""
    from fusiontransformer_amd import functional as spf
    from fusiontransformer_amd.trainer import fusion_losses
    g = load("losses_metric.npz")
    names = ("lidar_seg_logit", "img_seg_logit", "lidar_seg_logit2", "img_seg_logit2")
    label = torch.from_numpy(g["label"]).cuda()
    cw = torch.from_numpy(g["class_weights"]).cuda()
    lam = float(g["lambda_xm"])
    for dual in (True, False):
        preds = {k: torch.from_numpy(g[k]).cuda().requires_grad_(True) for k in names}
        ref = {k: torch.from_numpy(g[k]).double().requires_grad_(True) for k in names}
        c3 = torch.zeros((20, 20), dtype=torch.int64, device="cuda")
        c2 = torch.zeros_like(c3)
        l2, l3 = spf.fusion_loss(preds, label, cw, lam, dual, conf3d=c3, conf2d=c2)
        r2, r3 = fusion_losses(ref, label.cpu(), cw.cpu().double(), lam, dual)
        (l2 + l3).backward()
        (r2 + r3).backward()
        assert abs(l2.item() - r2.item()) < 2e-6 and abs(l3.item() - r3.item()) < 2e-6
        if dual:
            assert abs(l2.item() - float(g["loss_2d"])) < 2e-6 and abs(l3.item() - float(g["loss_3d"])) < 2e-6
        assert np.array_equal(c3.cpu().numpy(), g["mat3d"]) and np.array_equal(c2.cpu().numpy(), g["mat2d"])
        for k in names:
            if ref[k].grad is None:
                assert preds[k].grad is None
                continue
            np.testing.assert_allclose(preds[k].grad.cpu().numpy(), ref[k].grad.numpy(), rtol=1e-4, atol=1e-9)
    # unweighted form and a second accumulation into the same matrices
    preds = {k: torch.from_numpy(g[k]).cuda() for k in names}
    l2, l3 = spf.fusion_loss(preds, label, None, 0.0, True, conf3d=c3, conf2d=c2)
    r2, r3 = fusion_losses({k: v.cpu() for k, v in preds.items()}, label.cpu(), None, 0.0, True)
    assert abs(l2.item() - r2.item()) < 2e-6 and abs(l3.item() - r3.item()) < 2e-6
    assert np.array_equal(c3.cpu().numpy(), 2 * g["mat3d"])


@pytest.mark.gpu
def test_fused_loss_kernel_torchpack_mix_matches_reference_goldens():
    """ftx_fusion_loss_mix with ce_scale = 1 - lambda against the torchpack trainer's statements (values) and autograd through
    them (gradients), for configured / default class weights, dual / single head, lambda = 0."""
    from fusiontransformer_amd import functional as spf
    from fusiontransformer_amd.trainer import fusion_losses
    g, t = load("losses_metric.npz"), load("losses_torchpack.npz")
    names = ("lidar_seg_logit", "img_seg_logit", "lidar_seg_logit2", "img_seg_logit2")
    label = torch.from_numpy(g["label"]).cuda()
    for tag in ("cfgw", "defw", "single", "nolam"):
        w, lam, dual = torch.from_numpy(t[tag + "_weights"]), float(t[tag + "_lambda"]), bool(t[tag + "_dual"])
        preds = {k: torch.from_numpy(g[k]).cuda().requires_grad_(True) for k in names}
        ref = {k: torch.from_numpy(g[k]).double().requires_grad_(True) for k in names}
        l2, l3 = spf.fusion_loss(preds, label, w.cuda(), lam, dual, mix="torchpack")
        r2, r3 = fusion_losses(ref, label.cpu(), w.double(), lam, dual, mix="torchpack")
        (l2 + l3).backward()
        (r2 + r3).backward()
        assert abs(l2.item() - float(t[tag + "_loss_2d"])) < 2e-6 and abs(l3.item() - float(t[tag + "_loss_3d"])) < 2e-6, tag
        for k in names:
            if ref[k].grad is None:
                assert preds[k].grad is None or preds[k].grad.abs().max().item() == 0, (tag, k)
                continue
            np.testing.assert_allclose(preds[k].grad.cpu().numpy(), ref[k].grad.numpy(), rtol=1e-4, atol=1e-9)


@pytest.mark.gpu
def test_gpu_voxelisation_matches_reference_augment_and_scale():
    from fusiontransformer_amd.data.voxelize import points_to_voxels
    g = load("voxel_coords.npz")
    pts = g["points"]
    coords, keep = points_to_voxels(torch.from_numpy(pts).cuda())
    # reference: augment_and_scale_3d -> int64 -> range mask (golden), then first point per voxel in sorted-key order
    ci, valid = g["coords_int"], g["valid"]
    rows = np.nonzero(valid)[0]
    key = (ci[rows, 0] * 4096 + ci[rows, 1]) * 4096 + ci[rows, 2]
    _, inds = np.unique(key, return_index=True)
    assert np.array_equal(keep.cpu().numpy(), rows[inds])
    assert np.array_equal(coords.cpu().numpy(), ci[rows[inds]])


@pytest.mark.gpu
def test_eval_scatter_back_kernel_matches_reference_evaluator():
    from fusiontransformer_amd.evaluate import Evaluator, validate_batch
    g = load("eval_scatter_back.npz")
    nv, inv, gts = _split(g)
    names = ["c%d" % i for i in range(20)]
    ev = [Evaluator(names, labels=g["class_labels"]) for _ in range(3)]
    preds = {"lidar_seg_logit": torch.from_numpy(g["lidar_seg_logit"]).cuda(), "img_seg_logit": torch.from_numpy(g["img_seg_logit"]).cuda()}
    batch = {"orig_seg_label": gts, "inverse_map": inv, "sparse_orig_points_idx": [np.ones(n, dtype=bool) for n in nv]}
    out = validate_batch(preds, batch, g["class_labels"], *ev, want_preds=True)
    assert int(out["bad_index_flag"].item()) == 0
    assert np.array_equal(out["pred_3d"].cpu().numpy(), g["pred_3d"])
    assert np.array_equal(out["pred_2d"].cpu().numpy(), g["pred_2d"])
    # the ensemble adds two float32 softmaxes: allow argmax flips only where the two best sums tie to rounding
    pe = out["pred_ensemble"].cpu().numpy()
    assert (pe != g["pred_ensemble"]).mean() < 1e-3
    assert np.array_equal(ev[0].confusion_matrix, g["conf_3d"]) and np.array_equal(ev[1].confusion_matrix, g["conf_2d"])
    assert np.abs(ev[2].confusion_matrix - g["conf_ensemble"]).sum() <= 2 * (pe != g["pred_ensemble"]).sum()
    assert abs(ev[0].overall_iou - float(g["overall_iou_3d"])) < 1e-12 and abs(ev[0].overall_acc - float(g["overall_acc_3d"])) < 1e-12
    assert np.allclose(np.array(ev[0].class_iou), g["iou_3d"], equal_nan=True)
    # a second batch accumulates; an out-of-range inverse index raises the flag instead of reading out of bounds
    validate_batch(preds, batch, g["class_labels"], *ev)
    assert np.array_equal(ev[0].confusion_matrix, 2 * g["conf_3d"])
    bad = dict(batch, inverse_map=[inv[0] + 10 ** 6, inv[1]])
    assert int(validate_batch(preds, bad, g["class_labels"], *ev)["bad_index_flag"].item()) == 1
    # LiDAR-only model: no image logits
    ev1 = Evaluator(names, labels=g["class_labels"])
    validate_batch({"lidar_seg_logit": preds["lidar_seg_logit"]}, batch, g["class_labels"], evaluator_3d=ev1)
    assert np.array_equal(ev1.confusion_matrix, g["conf_3d"])


@pytest.mark.gpu
def test_projection_kernel_matches_reference_preprocess():
    from fusiontransformer_amd.data.preprocess import project_points, project_scan
    g = load("projection.npz")
    pts = torch.from_numpy(g["points"]).cuda()
    keep, rowcol = project_points(pts, torch.from_numpy(g["proj_matrix"]).cuda(), 1226, 370)
    keep = keep.cpu().numpy()
    assert np.array_equal(keep, g["keep_idx"])
    got = rowcol.cpu().numpy()[keep]
    assert np.array_equal(got.astype(np.int64), g["img_indices"])          # the pixel every point lifts from: exact
    assert np.array_equal(got, g["points_img"])                             # and the float coordinates bit for bit
    scan = np.concatenate([g["points"], np.linspace(0, 1, len(g["points"]), dtype=np.float32)[:, None]], 1)
    labels = (np.arange(len(scan)) % 20).astype(np.uint32) | np.uint32(7 << 16)   # upper half = instance id, dropped
    d = project_scan(scan, labels, g["proj_matrix"], (1226, 370))
    assert np.array_equal(d["points"], g["points"][g["keep_idx"]]) and np.array_equal(d["feats"], scan[g["keep_idx"]])
    assert d["seg_label"].dtype == np.int16 and np.array_equal(d["seg_label"], (np.arange(len(scan)) % 20)[g["keep_idx"]])
    assert np.array_equal(d["points_img"], g["points_img"])


def _aug_cases():
    g = load("voxel_coords_augmented.npz")
    for name in g["cases"]:
        for rep in range(2):
            tag = "%s_%d" % (name, rep)
            noisy_rot, flip_x, flip_y, rot_z, transl = (float(v) for v in g[tag + "_params"])
            yield tag, g["points"], int(g[tag + "_seed"]), dict(noisy_rot=noisy_rot, flip_x=flip_x, flip_y=flip_y, rot_z=rot_z, transl=bool(transl)), g[tag + "_coords_float"]


def test_augmentation_draws_and_cpu_restatement_match_the_reference_function():
    """draw_augmentation_3d consumes numpy.random in the reference's order (same seed -> same rotation matrix and offset), and the
    oracle's restatement of the arithmetic -- with the dot product's fused rounding written out -- reproduces the reference function's
    float coordinates bit for bit, for every augmentation mix of the fixture (data/utils/augmentation_3d.py:4-53)."""
    from fusiontransformer_amd.data.augment import draw_augmentation_3d
    n = 0
    for tag, points, seed, kw, want in _aug_cases():
        np.random.seed(seed)
        rot, u = draw_augmentation_3d(**kw)
        assert (rot is None) == (not (kw["noisy_rot"] > 0 or kw["flip_x"] > 0 or kw["flip_y"] > 0 or kw["rot_z"] > 0)) and (u is None) == (not kw["transl"])
        got = O.augment_and_scale_3d_np(points.copy(), 20, 4096, rot, u)
        assert got.dtype == np.float32 and np.array_equal(got, want), tag
        n += 1
    assert n == 10


@pytest.mark.gpu
def test_gpu_augmentation_matches_the_reference_function():
    """fusiontransformer_amd.data.augment.augment_and_scale_3d on the device (rotation by libftx's ftx_rotate_points) against the
    reference function's output: float coordinates, integer voxel coordinates and the in-range mask bit for bit."""
    import torch
    from fusiontransformer_amd.data.augment import augment_and_scale_3d, draw_augmentation_3d
    for tag, points, seed, kw, want in _aug_cases():
        np.random.seed(seed)
        rot, u = draw_augmentation_3d(**kw)
        got = augment_and_scale_3d(torch.from_numpy(points.copy()).cuda(), 20, 4096, rot, u)
        assert np.array_equal(got.cpu().numpy(), want), tag
        ci, wi = got.to(torch.int64).cpu().numpy(), want.astype(np.int64)                     # dataloader :220
        assert np.array_equal(ci, wi), tag
        assert np.array_equal((ci.min(1) >= 0) * (ci.max(1) < 4096), (wi.min(1) >= 0) * (wi.max(1) < 4096)), tag
