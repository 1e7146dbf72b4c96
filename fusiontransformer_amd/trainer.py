"""Train step of the fusion path: the counterpart of SemanticTrainer.train_step
(FusionTransformer/modules/SemanticTrainer.py:141-209).

Same loss: weighted CE x2 + lambda_xm * KL x2 in the additive form of
SemanticTrainer.py:158-178, Adam step.  Differences, all behaviour-preserving:
  * one backward of (loss_2d + loss_3d) instead of two backward calls: the image features enter
    the LiDAR branch detached (middle_fusion.py:102), so the two loss graphs are disjoint and
    the summed gradients are identical;
  * no `.item()` / `.cpu()` host syncs inside the step (the reference has 4+ per step);
    losses and the IoU confusion matrices stay on the device."""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F


def default_class_weights(num_classes, device=None):
    """The torchpack trainer's fallback when cfg.TRAIN.CLASS_WEIGHTS is empty (SemanticTorchpackTrainer.py:28-32): ones, class 0 ignored."""
    w = torch.ones(num_classes, device=device)
    w[0] = 0
    return w


def fusion_losses(preds, seg_label, class_weights, lambda_xm, dual_head, mix="additive"):
    """(loss_2d, loss_3d) exactly as SemanticTrainer.py:158-178 (mix="additive") or as the torchpack / DDP trainer's
    calc_loss, SemanticTorchpackTrainer.py:70-106 (mix="torchpack": (1-lambda)*CE + lambda*KL)."""
    seg_label = seg_label.long()
    loss_3d = F.cross_entropy(preds["lidar_seg_logit"], seg_label, weight=class_weights)
    loss_2d = F.cross_entropy(preds["img_seg_logit"], seg_label, weight=class_weights)
    if lambda_xm > 0:
        seg_logit_2d = preds["img_seg_logit2"] if dual_head else preds["img_seg_logit"]
        seg_logit_3d = preds["lidar_seg_logit2"] if dual_head else preds["lidar_seg_logit"]
        xm_loss_2d = F.kl_div(F.log_softmax(seg_logit_2d, dim=1), F.softmax(preds["lidar_seg_logit"].detach(), dim=1),
                              reduction="none").sum(1).mean()
        xm_loss_3d = F.kl_div(F.log_softmax(seg_logit_3d, dim=1), F.softmax(preds["img_seg_logit"].detach(), dim=1),
                              reduction="none").sum(1).mean()
        if mix == "torchpack":
            loss_2d = (1 - lambda_xm) * loss_2d + lambda_xm * xm_loss_2d
            loss_3d = (1 - lambda_xm) * loss_3d + lambda_xm * xm_loss_3d
        else:
            loss_2d = loss_2d + lambda_xm * xm_loss_2d
            loss_3d = loss_3d + lambda_xm * xm_loss_3d
    return loss_2d, loss_3d


def build_optimizer(cfg, model):
    """common/solver/build.py:7-20: getattr(torch.optim, TYPE)(params, lr, weight_decay)."""
    params = [p for p in model.parameters() if p.requires_grad]
    kwargs = dict(lr=cfg.OPTIMIZER.BASE_LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY)
    on_gpu = bool(params) and params[0].is_cuda
    if cfg.OPTIMIZER.TYPE == "Adam" and on_gpu and os.environ.get("FTX_ADAM", "1") != "0":
        from .optim import Adam      # torch.optim.Adam's rule and state layout, stepped by one libftx launch (csrc/ftx_optim.hip)
        return Adam(params, **kwargs)
    if cfg.OPTIMIZER.TYPE in ("Adam", "AdamW") and on_gpu:
        kwargs["fused"] = True  # same update rule, one multi-tensor kernel instead of ~10 passes over 108 M parameters
    return getattr(torch.optim, cfg.OPTIMIZER.TYPE)(params, **kwargs)


class TrainStep:
    def __init__(self, cfg, model, optimizer=None, metrics=None, grad_reducer=None, loss_mix="additive"):
        """loss_mix="additive": SemanticTrainer.train_step (the single-process trainer); "torchpack": the DDP trainer's mix and its
        default class weights (SemanticTorchpackTrainer.py:28-32,70-106) -- the one BASELINE configs[3] / [4] follow."""
        if loss_mix not in ("additive", "torchpack"):
            raise ValueError("loss_mix must be 'additive' or 'torchpack'")
        self.cfg, self.model, self.loss_mix = cfg, model, loss_mix
        self.optimizer = optimizer if optimizer is not None else build_optimizer(cfg, model)
        dev = next(model.parameters()).device
        cw = cfg.TRAIN.CLASS_WEIGHTS
        self.class_weights = torch.tensor(cw, dtype=torch.float32, device=dev) if len(cw) > 0 else None
        if self.class_weights is None and loss_mix == "torchpack":
            self.class_weights = default_class_weights(int(cfg.MODEL.NUM_CLASSES), dev)
        self.lambda_xm = float(cfg.TRAIN.FusionTransformer.lambda_xm)
        self.dual_head = bool(cfg.MODEL.DUAL_HEAD)
        self.metrics = metrics or ()
        self.grad_reducer = grad_reducer
        self.fused_loss = True
        self.prefetch_wait = False     # next_batch: start its index build without blocking this thread (SPVCNN.prepare(wait=False))
        self.prefetch_budget_ms = float(os.environ.get("FTX_PREFETCH_BUDGET_MS", "1.5"))   # ... then poll its host reads this long at most
        self._prefetch_misses = self._prefetch_pause = 0
        self.last = {}

    def reset_prefetch(self):
        """Forget what the self-pausing index prefetch learnt (call when the batch size or shape of the stream of batches changes)."""
        self._prefetch_misses = self._prefetch_pause = 0

    def __call__(self, data_batch, next_batch=None):
        """One training step on `data_batch`.  `next_batch` (optional): the batch of the FOLLOWING step.  Its index build (voxel sets of the
        five levels, kernel maps, point <-> voxel indices: ~230 small kernels and two host reads, models/_fusion_common.prepare_batch)
        is started on a third stream as soon as this step's backward has been issued, without blocking this thread; after the
        optimizer has been issued the build's reads are polled for at most `prefetch_budget_ms` and every part whose sizes have arrived
        is issued too; what is left is finished by the next forward.  The chain of small dependent kernels then runs BESIDE this step's
        backward instead of after it.  Same kernels and results either way (tests/test_model_gpu.py).  `prefetch_wait=True` builds
        everything with blocking reads (for a caller that owns an idle moment: a data-loading worker, an evaluation loop)."""
        ready = None
        if next_batch is not None and not self.prefetch_wait and self._prefetch_pause > 0:
            self._prefetch_pause -= 1        # the polls kept running out (large batches, see below): this step builds nothing ahead
            next_batch = None
        if next_batch is not None and torch.cuda.is_available():
            ready = torch.cuda.Event()       # whatever produced next_batch was queued before this point: its index build may start
            ready.record()                   # here and run beside this step, not behind it
        if self.grad_reducer is not None:
            self.grad_reducer.begin_step()   # zeroes the flat gradient buckets (p.grad are views into them)
        else:
            self.optimizer.zero_grad(set_to_none=True)    # first write of each gradient is a move, not fill + add
        preds = self.model(data_batch)
        logits = preds["lidar_seg_logit"]
        if self.fused_loss and logits.is_cuda and logits.shape[1] % 4 == 0 and logits.shape[1] <= 32:
            # one fused pass: CE x2 + KL x2 + their gradients + both SegIoU confusion matrices (libftx)
            conf = {"3d": None, "2d": None}
            for m in self.metrics:
                if m.mat is None:
                    m.mat = torch.zeros((m.num_classes, m.num_classes), dtype=torch.int64, device=logits.device)
                conf["3d" if "3d" in m.name else "2d"] = m.mat
            from . import functional as spf
            loss_2d, loss_3d = spf.fusion_loss(preds, data_batch["seg_label"], self.class_weights, self.lambda_xm, self.dual_head,
                                               conf3d=conf["3d"], conf2d=conf["2d"], mix=self.loss_mix)
        else:
            loss_2d, loss_3d = fusion_losses(preds, data_batch["seg_label"], self.class_weights, self.lambda_xm, self.dual_head, self.loss_mix)
            with torch.no_grad():
                for m in self.metrics:
                    m.update_dict(preds, data_batch)
        (loss_2d + loss_3d).backward()
        if self.grad_reducer is not None:
            self.grad_reducer.finish()       # the remaining gradient buckets go out before anything else is issued
        if next_batch is not None:
            from .models._fusion_common import prepare_batch
            prepare_batch(self.model, next_batch, ready=ready, wait=self.prefetch_wait)
        self.optimizer.step()
        if next_batch is not None and not self.prefetch_wait:
            # Small batches: the build's host reads arrive within a fraction of a millisecond, so a short bounded poll gets the whole
            # build issued before the next forward starts (batch 1: 14.9 -> 13.8 ms per step).  Large batches: they do not -- the small
            # index kernels sit behind the backward's big ones -- and both the poll (batch 4: +1.1 ms) and the early start (+0.3 ms)
            # cost more than they give: after two polls in a row that ran out, the next 256 steps build their index inside their own
            # forward, then the prefetch is tried again.
            from .models._fusion_common import advance_prepared
            complete = advance_prepared(next_batch, self.prefetch_budget_ms)
            self._prefetch_misses = 0 if complete else self._prefetch_misses + 1
            if self._prefetch_misses >= 2:
                self._prefetch_pause, self._prefetch_misses = 256, 0
        self.last = {"loss_2d": loss_2d.detach(), "loss_3d": loss_3d.detach()}
        return preds
