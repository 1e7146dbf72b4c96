"""LateFusionTransformer: mirror of FusionTransformer/models/late_fusion.py:5-58."""
from __future__ import annotations

import torch.nn as nn

from ._fusion_common import fused_outputs, heads, image_branch, lidar_preds, run_fusion
from .spvcnn import SPVCNN


class Net3DSeg(nn.Module):
    def __init__(self, num_classes, dual_head, backbone_3d_kwargs=dict()):
        super(Net3DSeg, self).__init__()
        self.backbone = SPVCNN(**backbone_3d_kwargs)
        heads(self, self.backbone.cs[-1], num_classes, dual_head)

    def forward(self, x):
        return lidar_preds(self, self.backbone(x))

    def forward_steps(self, x):
        feats = yield from self.backbone._backbone_steps(x)
        return lidar_preds(self, feats)


class LateFusionTransformer(nn.Module):
    def __init__(self, num_class, dual_head, backbone_3d_kwargs, backbone_2d_kwargs):
        super(LateFusionTransformer, self).__init__()
        self.dual_head = dual_head
        self.lidar_backbone = Net3DSeg(num_classes=num_class, dual_head=dual_head, backbone_3d_kwargs=backbone_3d_kwargs)
        self.image_backbone = image_branch(num_class, dual_head, backbone_2d_kwargs)

    def forward(self, data_dict):
        preds_lidar, preds_image = run_fusion(self, data_dict, lambda feats: self.lidar_backbone.forward_steps(data_dict["lidar"]),
                                              overlap=getattr(self, "overlap_branches", True))
        return fused_outputs(self.dual_head, preds_lidar, preds_image)
