"""MiddleFusionTransformer: mirror of FusionTransformer/models/middle_fusion.py:10-112."""
from __future__ import annotations

import torch.nn as nn

from ._fusion_common import fused_outputs, heads, image_branch, lidar_preds, run_fusion
from .spvcnn import SPVCNN, BatchNorm, _linear_bn_relu


class Net3DSeg(SPVCNN):
    def __init__(self, num_classes, dual_head, backbone_3d_kwargs=dict()):
        super(Net3DSeg, self).__init__(**backbone_3d_kwargs)
        self.middle_fusion_transform = nn.Sequential(nn.Linear(96, self.cs[4]), BatchNorm(self.cs[4]), nn.ReLU(True))
        heads(self, self.cs[-1], num_classes, dual_head)

    def _fuse(self, img_middle_feats):
        # z1.F = z1.F + point_transforms[0](z0.F) + middle_fusion_transform(img_middle_feats)  (middle_fusion.py:48)
        # img_middle_feats may be a lazy hand-off from the image stream: it is only touched at the fusion point
        def fuse():
            feats = img_middle_feats.get() if hasattr(img_middle_feats, "get") else img_middle_feats
            return _linear_bn_relu(self.middle_fusion_transform, feats)
        return fuse

    def backbone_forward_pass(self, x, img_middle_feats):
        return self._backbone(x, fuse_middle=self._fuse(img_middle_feats))

    def forward(self, x, img_middle_feats):
        return lidar_preds(self, self.backbone_forward_pass(x, img_middle_feats))

    def forward_steps(self, x, img_middle_feats):
        feats = yield from self._backbone_steps(x, fuse_middle=self._fuse(img_middle_feats))
        return lidar_preds(self, feats)


class MiddleFusionTransformer(nn.Module):
    def __init__(self, num_class, dual_head, backbone_3d_kwargs, backbone_2d_kwargs):
        super(MiddleFusionTransformer, self).__init__()
        self.dual_head = dual_head
        self.lidar_backbone = Net3DSeg(num_classes=num_class, dual_head=dual_head, backbone_3d_kwargs=backbone_3d_kwargs)
        self.image_backbone = image_branch(num_class, dual_head, backbone_2d_kwargs)

    def forward(self, data_dict):
        preds_lidar, preds_image = run_fusion(self, data_dict, lambda feats: self.lidar_backbone.forward_steps(data_dict["lidar"], feats),
                                              overlap=getattr(self, "overlap_branches", True))
        return fused_outputs(self.dual_head, preds_lidar, preds_image)
