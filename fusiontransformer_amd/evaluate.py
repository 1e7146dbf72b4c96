"""Validation-time mapping and metrics: the counterpart of FusionTransformer/data/utils/evaluate.py
(Evaluator) and of the per-batch body of data/utils/validate.py:62-120.

The reference moves the (N, 20) logits of every batch to the host, takes the argmax / softmax-sum there,
indexes through `inverse_map` with numpy, runs np.vectorize label maps and sklearn's confusion_matrix per
frame.  Here one kernel (`ftx_eval_scatter_back`) does all of it on the device; the confusion matrices stay
on the device until a property of the Evaluator is read."""
from __future__ import annotations

import numpy as np
import torch

from . import functional as spf


class Evaluator:
    """Same constructor and read-out API as the reference Evaluator (data/utils/evaluate.py:4-84)."""

    def __init__(self, class_names, labels=None, device="cuda"):
        self.class_names = tuple(class_names)
        self.num_classes = len(class_names)
        self.labels = np.arange(self.num_classes) if labels is None else np.array(labels)
        assert self.labels.shape[0] == self.num_classes
        self.device = torch.device(device)
        self.mat = torch.zeros((self.num_classes, self.num_classes), dtype=torch.int64, device=self.device)

    @property
    def confusion_matrix(self):
        return self.mat.cpu().numpy().astype(np.float64)

    @property
    def overall_acc(self):
        cm = self.confusion_matrix
        return np.sum(np.diag(cm)) / np.sum(cm)

    @property
    def overall_iou(self):
        class_iou = np.array(self.class_iou.copy())
        class_iou[np.isnan(class_iou)] = 0
        return np.mean(class_iou)

    @property
    def class_seg_acc(self):
        cm = self.confusion_matrix
        return [cm[i, i] / np.sum(cm[i]) for i in range(self.num_classes)]

    @property
    def class_iou(self):
        cm = self.confusion_matrix
        out = []
        for i in range(self.num_classes):
            tp, p, g = cm[i, i], cm[:, i].sum(), cm[i, :].sum()
            union = p + g - tp
            out.append(float("nan") if union == 0 else tp / union)
        return out

    def print_table(self):
        from tabulate import tabulate
        acc, iou, cm = self.class_seg_acc, self.class_iou, self.confusion_matrix
        table = [[name, acc[i] * 100, iou[i] * 100, int(cm[i].sum())] for i, name in enumerate(self.class_names)]
        return tabulate(table, headers=["Class", "Accuracy", "IOU", "Total"], tablefmt="psql", floatfmt=".2f")


def pack_inverse_maps(inverse_maps, points_per_frame, device):
    """list[B] of per-frame inverse maps -> one (sum M_b,) int64 tensor of global model-point rows (one H2D copy)."""
    offs = np.concatenate([[0], np.cumsum(points_per_frame)[:-1]]).astype(np.int64)
    packed = np.concatenate([np.asarray(im, dtype=np.int64) + o for im, o in zip(inverse_maps, offs)])
    return torch.from_numpy(packed).to(device, non_blocking=True)


def validate_batch(preds, data_batch, class_labels, evaluator_3d=None, evaluator_2d=None, evaluator_ensemble=None, want_preds=False):
    """The per-batch body of validate.py:62-120: `data_batch` carries `orig_seg_label` (list[B] of (M_b,) learning
    ids), `inverse_map` (list[B] of (M_b,)) and `sparse_orig_points_idx` (list[B] of bool masks, all True) as the
    reference's collate (data/collate.py) produces them.  Returns the per-original-point predictions (original
    label ids) when `want_preds`."""
    l3, l2 = preds.get("lidar_seg_logit"), preds.get("img_seg_logit")
    ref = l3 if l3 is not None else l2
    if "inverse_map_packed" in data_batch:
        # batch built on the device (data/voxelize.collate_device): the maps are already packed, nothing crosses PCIe
        inverse = data_batch["inverse_map_packed"]
        gt = data_batch["orig_seg_label_packed"]
        if inverse.shape[0] != gt.shape[0]:
            raise ValueError("validate_batch: packed inverse map and labels differ in length")
    else:
        pts = [int(np.sum(np.asarray(p))) for p in data_batch["sparse_orig_points_idx"]]
        for p, idx in zip(pts, data_batch["sparse_orig_points_idx"]):
            assert p == len(idx), "every voxel must carry a prediction (validate.py:87)"
        inverse = pack_inverse_maps(data_batch["inverse_map"], pts, ref.device)
        gt = torch.from_numpy(np.concatenate([np.asarray(g, dtype=np.int32) for g in data_batch["orig_seg_label"]])).to(ref.device, non_blocking=True)
    labels = torch.as_tensor(np.asarray(class_labels, dtype=np.int32))
    p3, p2, pe, bad = spf.eval_scatter_back(l3, l2, inverse, gt, labels,
                                            conf3d=None if evaluator_3d is None else evaluator_3d.mat,
                                            conf2d=None if evaluator_2d is None else evaluator_2d.mat,
                                            conf_ens=None if evaluator_ensemble is None else evaluator_ensemble.mat, want_preds=want_preds)
    return {"pred_3d": p3, "pred_2d": p2, "pred_ensemble": pe, "bad_index_flag": bad}
