"""ImageSegBilinear: mirror of FusionTransformer/models/image_models.py:23-36.

`ImageSeg` (the spatial-transformer variant, image_models_stn.py) is not used by any
fusion model and is out of scope."""
import torch.nn as nn

from .image_models_billinear import Net2DBillinear


class ImageSegBilinear(nn.Module):
    def __init__(self, num_classes, dual_head, backbone_2d_kwargs):
        super(ImageSegBilinear, self).__init__()
        self.image_backbone = Net2DBillinear(num_classes=num_classes, dual_head=dual_head, backbone_2d_kwargs=backbone_2d_kwargs)

    def forward(self, data_dict):
        preds_image = self.image_backbone(data_dict["img"], data_dict["img_indices"], lift_size=data_dict.get("lift_size"))
        return {"img_seg_logit": preds_image["img_seg_logit"]}
