"""Generates tests/golden/*.npz by RUNNING the reference's own importable code in the build
container (PYTHONPATH=/root/reference).  Run once here; the .npz files are committed, the
reference itself never travels.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is pinned, and by which reference code:
  upsample_index.npz   torch.nn.Upsample (the module BilinearModule instantiates,
                       models/image_models_billinear.py:17): nearest source index for every
                       (n_in -> n_out) pair the path uses.
  bilinear_lift.npz    reference BilinearModule.forward and Net2DBillinear.get_img_feats
                       (models/image_models_billinear.py:8-24,88-126) executed on seeded inputs.
                       Their module imports `timm` only to subclass / register the ViT
                       (models/transformers.py:1-8); empty placeholder modules carrying no
                       arithmetic satisfy those import lines, the classes exercised here are
                       pure torch.
  losses_metric.npz    weighted CE + KL exactly as the statements of
                       modules/SemanticTrainer.py:158-178 (that method cannot be called: it
                       lives in a class that needs wandb/torchsparse), and the reference's
                       SegIoU (models/metric.py:26-82) run on seeded logits.
  losses_torchpack.npz the torchpack / DDP trainer's loss mix and default class weights, statements of
                       modules/SemanticTorchpackTrainer.py:28-32,70-106, on the logits of losses_metric.npz.
  voxel_coords.npz     reference augment_and_scale_3d (data/utils/augmentation_3d.py:4-53) and
                       the int cast / range mask of semantic_kitti_dataloader.py:216-225.
  voxel_coords_augmented.npz  the same function with its augmentation branch on (noisy rotation, flips, rotation about z,
                       translation; data/utils/augmentation_3d.py:22-51), numpy.random seeded per case.
  projection.npz       reference DummyDataset.read_calib / select_points_in_frustum and the
                       projection statements of data/semantic_kitti/preprocess.py:54-116 on a
                       synthetic calib file + scan.
  eval_scatter_back.npz  reference map_sparse_to_org (data/utils/validate.py:10-11), Evaluator
                       (data/utils/evaluate.py:4-61) and the label inverse map built as in
                       semantic_kitti_dataloader.py:89-92 from semantic_kitti_label.yaml, driven by
                       the statements of validate.py:62-120 on seeded logits of two frames."""
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def upsample_index():
    tables = {}
    for n_in, n_out in [(24, 370), (24, 1226), (24, 384), (24, 1248), (24, 900), (24, 1600), (370, 384), (1226, 384), (900, 384), (1600, 384), (384, 384)]:
        src = torch.arange(n_in, dtype=torch.float32).view(1, 1, n_in, 1)
        up = torch.nn.Upsample((n_out, 1))(src)
        tables[f"{n_in}_{n_out}"] = up.view(-1).to(torch.int64).numpy()
    np.savez_compressed(os.path.join(OUT, "upsample_index.npz"), **tables)


def _placeholder_timm():
    """Import-line placeholders only: no arithmetic, never called by the code under test."""
    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    helpers = types.ModuleType("timm.models.helpers")
    vt = types.ModuleType("timm.models.vision_transformer")
    registry = types.ModuleType("timm.models.registry")
    helpers.overlay_external_default_cfg = lambda *a, **k: None
    vt.VisionTransformer = type("VisionTransformer", (torch.nn.Module,), {})
    vt.default_cfgs, vt.build_model_with_cfg, vt.checkpoint_filter_fn = {}, None, None
    registry.register_model = lambda f: f
    timm.models = models
    for name, mod in [("timm", timm), ("timm.models", models), ("timm.models.helpers", helpers),
                      ("timm.models.vision_transformer", vt), ("timm.models.registry", registry)]:
        sys.modules[name] = mod


def bilinear_lift():
    _placeholder_timm()
    from FusionTransformer.models.image_models_billinear import BilinearModule, Net2DBillinear
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    # (a) the down-sampler instance (3 -> 3, 370x1226 -> 384x384), train mode (batch statistics).
    # Inputs are stored quantised (int16 / 256, int8 / 32) to keep the fixtures small; the values
    # fed to the reference are exactly the de-quantised ones.
    down = BilinearModule(in_features=3, out_features=3, interpolation_output_size=(384, 384))
    down.train()
    img_q = rng.integers(-1000, 1000, size=(1, 3, 370, 1226)).astype(np.int16)
    img = torch.from_numpy(img_q.astype(np.float32) / 256.0)
    down_sd = {"down." + k: v.clone().numpy() for k, v in down.state_dict().items()}   # before the running-stat update
    down_out = down(img).detach().numpy()
    # (b) get_img_feats on a Net2DBillinear shell (no ViT needed for this method)
    up = BilinearModule(in_features=768, out_features=96, interpolation_output_size=(370, 1226))
    up.train()
    up_sd = {"up." + k: v.clone().numpy() for k, v in up.state_dict().items()}
    shell = Net2DBillinear.__new__(Net2DBillinear)
    torch.nn.Module.__init__(shell)
    shell.up = torch.nn.ModuleDict({"5": up})
    tok_q = rng.integers(-100, 100, size=(2, 576, 768)).astype(np.int8)
    tokens = torch.from_numpy(tok_q.astype(np.float32) / 32.0)
    idx = [np.stack([rng.integers(0, 370, 700), rng.integers(0, 1226, 700)], 1).astype(np.int64) for _ in range(2)]
    idx[0][:4] = [[0, 0], [369, 1225], [0, 1225], [369, 0]]
    feats = shell.get_img_feats(img_indices=idx, block_id="5", image_shape=(2, 3, 370, 1226), backbone_output={"5": tokens}).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "bilinear_lift.npz"), img_q=img_q, down_out=down_out, tok_q=tok_q, idx0=idx[0], idx1=idx[1],
                        feats=feats, down_running_mean_after=down.stem[2].running_mean.numpy(), down_running_var_after=down.stem[2].running_var.numpy(),
                        **down_sd, **up_sd)


def losses_metric():
    import torch.nn.functional as F
    from FusionTransformer.models.metric import SegIoU
    rng = np.random.default_rng(1)
    n = 4000
    preds = {k: torch.from_numpy(rng.standard_normal((n, 20)).astype(np.float32) * 2) for k in
             ("lidar_seg_logit", "img_seg_logit", "lidar_seg_logit2", "img_seg_logit2")}
    label = torch.from_numpy(rng.integers(0, 20, n).astype(np.int64))
    cw = torch.tensor([0., 1.58003993, 3.69774469, 3.2460013, 2.65342029, 2.61079801, 3.27744058, 3.48282471, 3.45874555, 1.,
                       2.07298878, 1.26831551, 2.65889542, 1.37436805, 1.4891881, 1.03083152, 2.25629999, 1.51838281, 2.51986332, 3.08564901])
    lam = 0.1
    # statements of SemanticTrainer.py:158-178 with DUAL_HEAD = True
    loss_3d = F.cross_entropy(preds['lidar_seg_logit'], label.long(), weight=cw)
    loss_2d = F.cross_entropy(preds['img_seg_logit'], label.long(), weight=cw)
    xm_loss_2d = F.kl_div(F.log_softmax(preds['img_seg_logit2'], dim=1), F.softmax(preds['lidar_seg_logit'].detach(), dim=1), reduction='none').sum(1).mean()
    xm_loss_3d = F.kl_div(F.log_softmax(preds['lidar_seg_logit2'], dim=1), F.softmax(preds['img_seg_logit'].detach(), dim=1), reduction='none').sum(1).mean()
    loss_2d = loss_2d + lam * xm_loss_2d
    loss_3d = loss_3d + lam * xm_loss_3d
    m3, m2 = SegIoU(20, name='seg_iou_3d'), SegIoU(20, name='seg_iou_2d')
    m3.update_dict(preds, {"seg_label": label})
    m2.update_dict(preds, {"seg_label": label})
    np.savez_compressed(os.path.join(OUT, "losses_metric.npz"), label=label.numpy(), class_weights=cw.numpy(), lambda_xm=lam,
                        loss_2d=loss_2d.numpy(), loss_3d=loss_3d.numpy(), mat3d=m3.mat.numpy(), mat2d=m2.mat.numpy(),
                        iou3d=m3.iou.numpy(), iou2d=m2.iou.numpy(), **{k: v.numpy() for k, v in preds.items()})


def losses_torchpack():
    """Statements of modules/SemanticTorchpackTrainer.py:28-32 (class weights) and :70-106 (calc_loss, USE_FUSION branch); the method
    itself lives in a torchpack Trainer subclass that cannot be imported here.  Same seeded logits as losses_metric."""
    import torch.nn.functional as F
    g = np.load(os.path.join(OUT, "losses_metric.npz"))
    preds = {k: torch.from_numpy(g[k]) for k in ("lidar_seg_logit", "img_seg_logit", "lidar_seg_logit2", "img_seg_logit2")}
    label = torch.from_numpy(g["label"])
    save = {}
    for tag, weights, lam, dual in (("cfgw", torch.from_numpy(g["class_weights"]), 0.1, True), ("defw", None, 0.1, True),
                                    ("single", torch.from_numpy(g["class_weights"]), 0.25, False), ("nolam", None, 0.0, True)):
        if weights is not None:                       # SemanticTorchpackTrainer.py:28-29
            class_weights = weights
        else:                                         # :30-32
            class_weights = torch.ones(20)
            class_weights[0] = 0
        loss_3d = F.cross_entropy(preds['lidar_seg_logit'], label.long(), weight=class_weights)
        loss_2d = F.cross_entropy(preds['img_seg_logit'], label.long(), weight=class_weights)
        if lam > 0:
            seg_logit_2d = preds['img_seg_logit2'] if dual else preds['img_seg_logit']
            seg_logit_3d = preds['lidar_seg_logit2'] if dual else preds['lidar_seg_logit']
            xm_loss_2d = F.kl_div(F.log_softmax(seg_logit_2d, dim=1), F.softmax(preds['lidar_seg_logit'].detach(), dim=1), reduction='none').sum(1).mean()
            xm_loss_3d = F.kl_div(F.log_softmax(seg_logit_3d, dim=1), F.softmax(preds['img_seg_logit'].detach(), dim=1), reduction='none').sum(1).mean()
            loss_2d = (1 - lam) * loss_2d + (lam) * xm_loss_2d
            loss_3d = (1 - lam) * loss_3d + (lam) * xm_loss_3d
        save.update({tag + "_loss_2d": loss_2d.numpy(), tag + "_loss_3d": loss_3d.numpy(), tag + "_lambda": lam, tag + "_dual": dual,
                     tag + "_weights": class_weights.numpy()})
    np.savez_compressed(os.path.join(OUT, "losses_torchpack.npz"), **save)


def voxel_coords():
    from FusionTransformer.data.utils.augmentation_3d import augment_and_scale_3d
    rng = np.random.default_rng(2)
    points = (rng.uniform(-1, 1, size=(5000, 3)) * np.array([60, 40, 3])).astype(np.float32)
    points[:, 0] = np.abs(points[:, 0])
    coords = augment_and_scale_3d(points, 20, 4096, noisy_rot=0.0, flip_y=0.0, rot_z=0.0, transl=False)
    coords_i = coords.astype(np.int64)
    valid = (coords_i.min(1) >= 0) * (coords_i.max(1) < 4096)
    np.savez_compressed(os.path.join(OUT, "voxel_coords.npz"), points=points, coords_float=coords, coords_int=coords_i, valid=valid)


def voxel_coords_augmented():
    """The augmentation branch of the reference function (noisy rotation, flips, z rotation, translation), seeded through numpy.random
    as the dataloader leaves it: points, parameters and seed in, float / integer coordinates and the in-range mask out."""
    from FusionTransformer.data.utils.augmentation_3d import augment_and_scale_3d
    rng = np.random.default_rng(12)
    points = (rng.uniform(-1, 1, size=(2500, 3)) * np.array([60, 40, 3])).astype(np.float32)
    points[:, 0] = np.abs(points[:, 0])
    cases = {
        "all": dict(noisy_rot=0.1, flip_x=0.5, flip_y=0.5, rot_z=2 * np.pi, transl=True),      # xmuda-style settings, everything on
        "kitti": dict(noisy_rot=0.1, flip_x=0.0, flip_y=0.5, rot_z=2 * np.pi, transl=True),    # the commented values of the fork's config
        "flip_only": dict(noisy_rot=0.0, flip_x=0.5, flip_y=0.0, rot_z=0.0, transl=False),
        "rot_only": dict(noisy_rot=0.0, flip_x=0.0, flip_y=0.0, rot_z=2 * np.pi, transl=False),
        "transl_only": dict(noisy_rot=0.0, flip_x=0.0, flip_y=0.0, rot_z=0.0, transl=True),
    }
    save = {"points": points, "cases": np.array(list(cases.keys()))}
    for i, (name, kw) in enumerate(cases.items()):
        for rep in range(2):
            seed = 100 + 10 * i + rep
            np.random.seed(seed)
            coords = augment_and_scale_3d(points.copy(), 20, 4096, **kw)
            tag = "%s_%d" % (name, rep)      # the int cast and the range mask (dataloader :220-225) are applied by the tests
            save.update({tag + "_seed": seed, tag + "_params": np.array([kw["noisy_rot"], kw["flip_x"], kw["flip_y"], kw["rot_z"], float(kw["transl"])]),
                         tag + "_coords_float": coords})
    np.savez_compressed(os.path.join(OUT, "voxel_coords_augmented.npz"), **save)


def projection():
    from FusionTransformer.data.semantic_kitti.preprocess import DummyDataset
    rng = np.random.default_rng(3)
    P2 = np.array([[718.856, 0.0, 607.1928, 45.38], [0.0, 718.856, 185.2157, -0.1130887], [0.0, 0.0, 1.0, 0.003779761]])
    Tr = np.array([[4.2768028e-04, -9.9996725e-01, -8.0844917e-03, -1.1984599e-02], [-7.2106265e-03, 8.0811985e-03, -9.9994132e-01, -5.4039847e-02],
                   [9.9997386e-01, 4.8594858e-04, -7.2069002e-03, -2.9219686e-01]])
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "calib.txt")
        with open(path, "w") as f:
            for name, m in (("P0", P2), ("P1", P2), ("P2", P2), ("P3", P2), ("Tr", Tr)):
                f.write(name + ": " + " ".join("%.12e" % v for v in m.reshape(-1)) + "\n")
        calib = DummyDataset.read_calib(path)
    proj_matrix = (calib['P2'] @ calib['Tr']).astype(np.float32)  # preprocess.py:32-33
    points = (rng.uniform(-1, 1, size=(6000, 3)) * np.array([50, 30, 3])).astype(np.float32)
    # statements of preprocess.py:108-116
    keep_idx = points[:, 0] > 0
    points_hcoords = np.concatenate([points[keep_idx], np.ones([keep_idx.sum(), 1], dtype=np.float32)], axis=1)
    img_points = (proj_matrix @ points_hcoords.T).T
    img_points = img_points[:, :2] / np.expand_dims(img_points[:, 2], axis=1)
    keep_idx_img_pts = DummyDataset.select_points_in_frustum(img_points, 0, 0, 1226, 370)
    keep_idx[keep_idx] = keep_idx_img_pts
    img_points = np.fliplr(img_points)
    np.savez_compressed(os.path.join(OUT, "projection.npz"), points=points, proj_matrix=proj_matrix, keep_idx=keep_idx,
                        points_img=img_points[keep_idx_img_pts], img_indices=img_points[keep_idx_img_pts].astype(np.int64))


def eval_scatter_back():
    import yaml
    import torch.nn.functional as F
    from FusionTransformer.data.utils.evaluate import Evaluator
    from FusionTransformer.data.utils.validate import map_sparse_to_org
    with open(os.path.join(REF, "FusionTransformer/data/semantic_kitti/semantic_kitti_label.yaml")) as f:
        cfg = yaml.safe_load(f)
    # semantic_kitti_dataloader.py:89-92
    class_names = [cfg["labels"][k] for k in cfg["learning_map_inv"].values()]
    class_labels = list(cfg["learning_map_inv"].copy().values())
    map_inverse_label = np.vectorize(lambda learning_label: cfg["learning_map_inv"][learning_label])
    rng = np.random.default_rng(5)
    n_vox = [1500, 1100]                       # voxels (= model points) per frame
    n_org = [4000, 3100]                       # original points per frame
    n = sum(n_vox)
    preds = {k: torch.from_numpy((rng.standard_normal((n, 20)) * 2).astype(np.float32)) for k in ("lidar_seg_logit", "img_seg_logit")}
    inverse_map = [rng.integers(0, nv, no).astype(np.int64) for nv, no in zip(n_vox, n_org)]
    orig_seg_label = [rng.integers(0, 20, no).astype(np.int64) for no in n_org]
    points_idx = [np.ones(nv, dtype=bool) for nv in n_vox]
    # statements of validate.py:62-120 with USE_FUSION
    pred_label_voxel_3d = preds['lidar_seg_logit'].argmax(1).cpu().numpy()
    pred_label_voxel_2d = preds['img_seg_logit'].argmax(1).cpu().numpy()
    probs_2d = F.softmax(preds['img_seg_logit'], dim=1)
    probs_3d = F.softmax(preds['lidar_seg_logit'], dim=1)
    pred_label_voxel_ensemble = (probs_2d + probs_3d).argmax(1).cpu().numpy()
    evaluator_3d, evaluator_2d, evaluator_ensemble = (Evaluator(class_names, labels=class_labels) for _ in range(3))
    out3, out2, oute = [], [], []
    left_idx = 0
    for batch_ind in range(len(orig_seg_label)):
        curr_points_idx = points_idx[batch_ind]
        assert np.all(curr_points_idx)
        curr_inverse_map = inverse_map[batch_ind]
        curr_seg_label = orig_seg_label[batch_ind].copy()
        right_idx = left_idx + curr_points_idx.sum()
        pred_label_3d = map_sparse_to_org(pred_label_voxel_3d[left_idx:right_idx], curr_inverse_map)
        pred_label_2d = map_sparse_to_org(pred_label_voxel_2d[left_idx:right_idx], curr_inverse_map)
        pred_label_ensemble = map_sparse_to_org(pred_label_voxel_ensemble[left_idx:right_idx], curr_inverse_map)
        curr_seg_label = map_inverse_label(curr_seg_label)
        pred_label_3d = map_inverse_label(pred_label_3d)
        pred_label_2d = map_inverse_label(pred_label_2d)
        pred_label_ensemble = map_inverse_label(pred_label_ensemble)
        out3.append(pred_label_3d.copy()); out2.append(pred_label_2d.copy()); oute.append(pred_label_ensemble.copy())
        evaluator_3d.update(pred_label_3d, curr_seg_label.copy())
        evaluator_2d.update(pred_label_2d, curr_seg_label.copy())
        evaluator_ensemble.update(pred_label_ensemble, curr_seg_label.copy())
        left_idx = right_idx
    save = dict(lidar_seg_logit=preds["lidar_seg_logit"].numpy(), img_seg_logit=preds["img_seg_logit"].numpy(), n_vox=np.array(n_vox), n_org=np.array(n_org),
                inverse_map=np.concatenate(inverse_map), orig_seg_label=np.concatenate(orig_seg_label), class_labels=np.array(class_labels),
                pred_3d=np.concatenate(out3), pred_2d=np.concatenate(out2), pred_ensemble=np.concatenate(oute),
                conf_3d=evaluator_3d.confusion_matrix, conf_2d=evaluator_2d.confusion_matrix, conf_ensemble=evaluator_ensemble.confusion_matrix,
                iou_3d=np.array(evaluator_3d.class_iou), overall_iou_3d=evaluator_3d.overall_iou, overall_acc_3d=evaluator_3d.overall_acc,
                overall_iou_ensemble=evaluator_ensemble.overall_iou)
    np.savez_compressed(os.path.join(OUT, "eval_scatter_back.npz"), **save)


if __name__ == "__main__":
    if len(sys.argv) > 1:          # regenerate only the named fixtures
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    upsample_index()
    bilinear_lift()
    losses_metric()
    losses_torchpack()
    voxel_coords()
    voxel_coords_augmented()
    projection()
    eval_scatter_back()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
