"""LidarSeg: mirror of FusionTransformer/models/lidar_model.py:4-21."""
import torch.nn as nn

from .. import functional as spf

from .spvcnn import SPVCNN


class LidarSeg(nn.Module):
    def __init__(self, num_classes, backbone_3d_kwargs):
        super(LidarSeg, self).__init__()
        self.backbone = SPVCNN(**backbone_3d_kwargs)
        self.linear = nn.Linear(self.backbone.cs[-1], num_classes)

    def forward(self, data_dict):
        feats = self.backbone(data_dict["lidar"])
        return {"lidar_seg_logit": spf.linear(feats, self.linear.weight, self.linear.bias)}
