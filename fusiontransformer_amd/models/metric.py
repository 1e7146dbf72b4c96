"""SegIoU: mirror of FusionTransformer/models/metric.py:26-82, accumulated on the device
(the reference copies both (sum N, 20) logit tensors to the host every step, metric.py:43-44)."""
import torch


class SegIoU(object):
    def __init__(self, num_classes, ignore_index=0, name="seg_iou"):
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        self.mat = None
        self.name = name

    def update_dict(self, preds, labels):
        if "3d" in self.name:
            seg_logit = preds["lidar_seg_logit"]
        if "2d" in self.name:
            seg_logit = preds["img_seg_logit"]
        seg_label = labels["seg_label"].detach().long().to(seg_logit.device)
        pred_label = seg_logit.detach().argmax(1)
        n = self.num_classes
        # points whose label is the ignored class, or outside [0, n) (e.g. -100), are not counted -- as ftx_fusion_loss does
        mask = (seg_label != self.ignore_index) & (seg_label >= 0) & (seg_label < n)
        with torch.no_grad():
            if self.mat is None:
                self.mat = seg_label.new_zeros((n, n))
            # same counts as the reference's bincount over the masked points, without its two host
            # synchronisations (boolean-mask indexing and bincount both need a size from the device)
            inds = (n * seg_label + pred_label).clamp_(0, n * n - 1)
            self.mat.view(-1).index_add_(0, inds, mask.to(self.mat.dtype))

    def reset(self):
        self.mat = None

    @property
    def iou(self):
        h = self.mat.float()
        return torch.diag(h) / (h.sum(1) + h.sum(0) - torch.diag(h))

    @property
    def global_avg(self):
        return self.iou.mean().item()

    @property
    def avg(self):
        return self.global_avg

    def __str__(self):
        return "{iou:.4f}".format(iou=self.iou.mean().item())

    @property
    def summary_str(self):
        return str(self)
