"""Shared pieces of the three fusion models (the reference repeats them per file)."""
from __future__ import annotations

import torch.nn as nn

from .. import functional as spf

from .image_models_billinear import Net2DBillinear


def heads(module, in_channels, num_classes, dual_head):
    module.linear = nn.Linear(in_channels, num_classes)
    module.dual_head = dual_head
    if dual_head:
        module.linear2 = nn.Linear(in_channels, num_classes)


def lidar_preds(module, feats):
    preds = {"lidar_feats": feats, "lidar_seg_logit": spf.linear(feats, module.linear.weight, module.linear.bias)}
    if module.dual_head:
        preds["lidar_seg_logit2"] = spf.linear(feats, module.linear2.weight, module.linear2.bias)
    return preds


def fused_outputs(dual_head, preds_lidar, preds_image):
    out = {"lidar_seg_logit": preds_lidar["lidar_seg_logit"], "img_seg_logit": preds_image["img_seg_logit"]}
    if dual_head:
        out.update({"lidar_seg_logit2": preds_lidar["lidar_seg_logit2"], "img_seg_logit2": preds_image["img_seg_logit2"]})
    return out


def image_branch(num_class, dual_head, backbone_2d_kwargs):
    return Net2DBillinear(num_classes=num_class, dual_head=dual_head, backbone_2d_kwargs=backbone_2d_kwargs)
