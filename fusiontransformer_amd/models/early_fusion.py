"""EarlyFusionTransformer: mirror of FusionTransformer/models/early_fusion.py:9-114."""
from __future__ import annotations

import torch.nn as nn

from ._fusion_common import fused_outputs, heads, image_branch, lidar_preds, run_fusion
from .spvcnn import SPVCNN, BatchNorm, _linear_bn_relu


class Net3DSeg(SPVCNN):
    def __init__(self, num_classes, dual_head, backbone_3d_kwargs=dict()):
        super(Net3DSeg, self).__init__(**backbone_3d_kwargs)
        self.early_fusion_transform = nn.Sequential(nn.Linear(96, 32), BatchNorm(32), nn.ReLU(True))
        heads(self, self.cs[-1], num_classes, dual_head)

    def _fuse(self, img_early_feats):
        # z0.F = z0.F + early_fusion_transform(img_early_feats)  (early_fusion.py:39)
        def fuse():
            feats = img_early_feats.get() if hasattr(img_early_feats, "get") else img_early_feats
            return _linear_bn_relu(self.early_fusion_transform, feats)
        return fuse

    def backbone_forward_pass(self, x, img_early_feats):
        return self._backbone(x, fuse_early=self._fuse(img_early_feats))

    def forward(self, x, img_early_feats):
        return lidar_preds(self, self.backbone_forward_pass(x, img_early_feats))

    def forward_steps(self, x, img_early_feats):
        feats = yield from self._backbone_steps(x, fuse_early=self._fuse(img_early_feats))
        return lidar_preds(self, feats)


class EarlyFusionTransformer(nn.Module):
    def __init__(self, num_class, dual_head, backbone_3d_kwargs, backbone_2d_kwargs):
        super(EarlyFusionTransformer, self).__init__()
        self.dual_head = dual_head
        self.lidar_backbone = Net3DSeg(num_classes=num_class, dual_head=dual_head, backbone_3d_kwargs=backbone_3d_kwargs)
        self.image_backbone = image_branch(num_class, dual_head, backbone_2d_kwargs)

    def forward(self, data_dict):
        # with middle_feat_block_number = 0 the "middle" tap is the early one (early_fusion.py:101-105)
        preds_lidar, preds_image = run_fusion(self, data_dict, lambda feats: self.lidar_backbone.forward_steps(data_dict["lidar"], feats),
                                              overlap=getattr(self, "overlap_branches", True))
        return fused_outputs(self.dual_head, preds_lidar, preds_image)
