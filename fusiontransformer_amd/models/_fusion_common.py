"""Shared pieces of the three fusion models (the reference repeats them per file)."""
from __future__ import annotations

import torch
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


class _Lazy:
    """Image features handed from the image stream to the LiDAR stream: `get()` makes the calling
    stream wait for the producer's event (not for the whole image branch)."""

    def __init__(self):
        self.feats, self.event = None, None

    def set(self, feats):
        self.feats = feats.detach()          # middle_fusion.py:102 / early_fusion.py:105
        if feats.is_cuda:
            self.event = torch.cuda.Event()
            self.event.record()

    def get(self):
        if self.feats is None:
            raise RuntimeError("the image branch did not produce the fused tap (check *_feat_block_number)")
        if self.event is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self.event)
            self.feats.record_stream(cur)
        return self.feats


_STREAMS = {}


def _branch_streams(device):
    key = (device.type, device.index)
    if key not in _STREAMS:
        _STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    return _STREAMS[key]


def bump_batchnorm_counters(model):
    """num_batches_tracked += 1 for every BatchNorm of the model in ONE multi-tensor launch (the
    per-module add_ is ~60 tiny kernels per step).  Every BatchNorm of these models runs exactly once
    per training forward, so the counters end up exactly as the per-module increments leave them."""
    bns = getattr(model, "_ftx_bns", None)
    if bns is None:
        bns = [m for m in model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.track_running_stats and m.num_batches_tracked is not None]
        for m in bns:
            m._nbt_external = True
        object.__setattr__(model, "_ftx_bns", bns)
    live = [m.num_batches_tracked for m in bns if m.training]
    if live:
        torch._foreach_add_(live, 1)


def prepare_batch(model, data_dict, ready=None, wait=True):
    """Build the coordinate structures of `data_dict` (voxel hash, the five levels' kernel maps, point <-> voxel indices) ahead of
    the forward that will receive it, on a stream of their own (SPVCNN.prepare).  The forward then finds them on
    data_dict["lidar"]; a batch that was not prepared builds them inside the forward as before.  `ready`: an event after which the
    batch's tensors are valid (default: everything queued on the current stream so far); `wait=False`: issue the build up to its first
    host read and return without blocking (SPVCNN.prepare).  Returns data_dict."""
    lb = getattr(model, "lidar_backbone", None)
    lidar = data_dict.get("lidar") if isinstance(data_dict, dict) else None
    if lb is not None and lidar is not None and hasattr(lb, "prepare"):
        lb.prepare(lidar, ready=ready, wait=wait)
    return data_dict


def advance_prepared(data_dict, budget_ms=0.0):
    """Push an index build started with prepare_batch(wait=False) forward: whenever the host read it is parked on has arrived, issue its
    next part; give up after `budget_ms` of polling (0: look once, never wait).  The wait is bounded on purpose: at small batches the
    reads arrive within a fraction of a millisecond and the whole build is then issued before the next forward starts; at large
    batches the build's small kernels sit behind the backward's big ones, and a thread that waited for them would have nothing queued
    for the GPU when they finally ran (measured: -3 ms per step at batch 4 with an unbounded wait).  Returns True when the build is complete."""
    import time
    lidar = data_dict.get("lidar") if isinstance(data_dict, dict) else None
    pending = getattr(lidar, "prepared", None)
    if pending is None or not hasattr(pending, "step"):
        return pending is not None
    deadline = time.perf_counter() + budget_ms * 1e-3
    while True:
        if pending.ready():
            if pending.step():
                lidar.prepared = pending.done
                return True
        elif time.perf_counter() >= deadline:
            return False


def _drain(steps):
    while True:
        try:
            next(steps)
        except StopIteration as done:
            return done.value


def run_fusion(model, data_dict, lidar_steps, overlap=True):
    """Runs the image branch and the LiDAR branch of a fusion model.

    `lidar_steps(lazy_feats)` returns a generator that issues the LiDAR branch stage by stage and
    returns its prediction dict.  On the GPU the two branches go to two HIP streams and their kernel
    launches are INTERLEAVED: after every ViT block the scheduler issues one LiDAR stage.  The ViT is a
    chain of large dense GEMMs, the SPVCNN a long chain of small gather / scatter kernels and
    host-synchronising index builds; they share nothing until the fusion add (one event), so they
    overlap -- in the forward and, because autograd replays nodes in reverse creation order on their
    forward streams, in the backward as well.  Interleaving the issue matters: issuing one branch
    completely before the other leaves the second stream empty for that long."""
    img = data_dict["img"]
    lazy = _Lazy()
    if model.training:
        bump_batchnorm_counters(model)
    if not (overlap and img.is_cuda):
        preds_image = model.image_backbone(img=img, img_indices=data_dict["img_indices"], on_middle=lazy.set, lift_size=data_dict.get("lift_size"))
        return _drain(lidar_steps(lazy)), preds_image
    cur = torch.cuda.current_stream()
    s_img, s_lid = _branch_streams(img.device)
    s_img.wait_stream(cur)
    s_lid.wait_stream(cur)
    gen = lidar_steps(lazy)
    state = {"done": False, "blocked": False, "preds": None}

    def pump():
        """Issue one more LiDAR stage (unless it waits for image features that do not exist yet)."""
        if state["done"]:
            return
        if state["blocked"]:
            if lazy.feats is None:
                return
            state["blocked"] = False
        with torch.cuda.stream(s_lid):
            try:
                token = next(gen)
            except StopIteration as done:
                state["preds"], state["done"] = done.value, True
                return
        if isinstance(token, str) and token.startswith("need_") and lazy.feats is None:
            state["blocked"] = True

    with torch.cuda.stream(s_img):
        preds_image = model.image_backbone(img=img, img_indices=data_dict["img_indices"], on_middle=lazy.set, on_step=pump,
                                           lift_size=data_dict.get("lift_size"))
    while not state["done"]:
        if state["blocked"] and lazy.feats is None:
            raise RuntimeError("the LiDAR branch needs image features the image branch never produced")
        pump()
    preds_lidar = state["preds"]
    cur.wait_stream(s_img)
    cur.wait_stream(s_lid)
    for d in (preds_image, preds_lidar):
        for v in d.values():
            if torch.is_tensor(v):
                v.record_stream(cur)
    return preds_lidar, preds_image
