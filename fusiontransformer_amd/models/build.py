"""build_model(cfg): mirror of FusionTransformer/models/build.py:9-88 (same dispatch on
cfg.MODEL.USE_FUSION / USE_LIDAR / USE_IMAGE and cfg.MODEL.TYPE, same return tuples)."""
from .early_fusion import EarlyFusionTransformer
from .image_models import ImageSegBilinear
from .late_fusion import LateFusionTransformer
from .lidar_model import LidarSeg
from .metric import SegIoU
from .middle_fusion import MiddleFusionTransformer


def build_metrics(cfg):
    train_3d_metric = SegIoU(num_classes=cfg.MODEL.NUM_CLASSES, name="seg_iou_3d")
    train_2d_metric = SegIoU(num_classes=cfg.MODEL.NUM_CLASSES, name="seg_iou_2d")
    return train_2d_metric, train_3d_metric


def _fusion(cls, cfg):
    train_2d_metric, train_3d_metric = build_metrics(cfg)
    model = cls(num_class=cfg.MODEL.NUM_CLASSES, dual_head=cfg.MODEL.DUAL_HEAD, backbone_2d_kwargs=cfg.MODEL, backbone_3d_kwargs=cfg.MODEL)
    return model, train_2d_metric, train_3d_metric


def build_late_fusion_model(cfg):
    return _fusion(LateFusionTransformer, cfg)


def build_middle_fusion_model(cfg):
    return _fusion(MiddleFusionTransformer, cfg)


def build_early_fusion_model(cfg):
    return _fusion(EarlyFusionTransformer, cfg)


def build_lidar_model(cfg):
    train_3d_metric = SegIoU(num_classes=cfg.MODEL.NUM_CLASSES, name="seg_iou_3d")
    return LidarSeg(num_classes=cfg.MODEL.NUM_CLASSES, backbone_3d_kwargs=cfg.MODEL), train_3d_metric


def build_image_bilinear_model(cfg):
    train_2d_metric = SegIoU(num_classes=cfg.MODEL.NUM_CLASSES, name="seg_iou_2d")
    return ImageSegBilinear(num_classes=cfg.MODEL.NUM_CLASSES, dual_head=cfg.MODEL.DUAL_HEAD, backbone_2d_kwargs=cfg.MODEL), train_2d_metric


def build_model(cfg):
    if cfg.MODEL.USE_FUSION:
        if cfg.MODEL.TYPE == "LateFusionTransformer":
            return build_late_fusion_model(cfg=cfg)
        if cfg.MODEL.TYPE == "MiddleFusionTransformer":
            return build_middle_fusion_model(cfg=cfg)
        if cfg.MODEL.TYPE == "EarlyFusionTransformer":
            return build_early_fusion_model(cfg)
    elif cfg.MODEL.USE_LIDAR:
        if cfg.MODEL.TYPE == "LidarSeg":
            return build_lidar_model(cfg)
    elif cfg.MODEL.USE_IMAGE:
        if cfg.MODEL.TYPE == "ImageSegBilinear":
            return build_image_bilinear_model(cfg)
        if cfg.MODEL.TYPE == "ImageSeg":
            raise NotImplementedError("ImageSeg (spatial-transformer variant) is out of scope: no fusion model uses it")
    raise ValueError("unsupported MODEL configuration: TYPE=%s" % cfg.MODEL.TYPE)
