"""Image branch + 2D->3D lift.

Mirror of FusionTransformer/models/image_models_billinear.py:8-155
(`BilinearModule`, `Net2DBillinear`), same attribute names / state_dict keys.

What changes on MI355X: the reference up-samples each tapped block's 96-channel
24x24 map to 370x1226 (174 MB fp32 per frame per tap, written in the forward
and again in the backward) and then picks ~20k pixels out of it.  Here
Conv1x1 -> ReLU -> BN run on the 24x24 token grid (BN statistics are taken
before the upsample in the reference too, image_models_billinear.py:20-22), and
`lift_gather` reads the nearest source cell directly for each point: the
up-sampled map never exists."""
from __future__ import annotations

import os

from collections import OrderedDict
from typing import Dict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as spf
from .transformers import image_2d_distilled_transformer

__all__ = ["BilinearModule", "Net2DBillinear", "pack_img_indices"]


def pack_img_indices(img_indices, device):
    """list[B] of (N_i,2) int64 (row,col) host arrays (data/collate.py:50,73) -> one
    (sum N,2) int64 device tensor + (sum N,) int32 frame index: ONE H2D per batch
    instead of the reference's per-frame implicit copies (image_models_billinear.py:118-123).
    An already packed (idx, frame) pair of device tensors is passed through."""
    if isinstance(img_indices, tuple) and len(img_indices) == 2 and torch.is_tensor(img_indices[0]):
        return img_indices
    arrs = [np.asarray(a.cpu() if torch.is_tensor(a) else a, dtype=np.int64).reshape(-1, 2) for a in img_indices]
    idx = torch.from_numpy(np.ascontiguousarray(np.concatenate(arrs, 0)))
    frame = torch.from_numpy(np.concatenate([np.full((a.shape[0],), i, dtype=np.int32) for i, a in enumerate(arrs)]))
    return idx.to(device, non_blocking=True), frame.to(device, non_blocking=True)


class BilinearModule(nn.Module):
    """Conv1x1 -> ReLU -> BatchNorm2d -> nn.Upsample(size) (nearest, despite the name)."""

    def __init__(self, in_features, out_features, interpolation_output_size):
        super().__init__()
        self.stem = nn.Sequential(
            nn.Conv2d(in_channels=in_features, out_channels=out_features, kernel_size=1),
            nn.ReLU(True),
            nn.BatchNorm2d(out_features))
        if isinstance(interpolation_output_size, str):     # "image": per-batch lift size, resolved in Net2DBillinear
            interpolation_output_size = (370, 1226)        # the reference literal (image_models_billinear.py:74,77)
        self.up = nn.Upsample(interpolation_output_size)   # kept for parity of the module tree; executed by libftx
        self.size = tuple(interpolation_output_size)

    def forward(self, x):
        # stem = Conv1x1 -> ReLU -> BN2d on the full-resolution map (the batch statistics need
        # every pixel), then the nearest pick
        conv, bn = self.stem[0], self.stem[2]
        if conv.in_channels == 3 and conv.out_channels == 3 and (bn.training or not torch.is_grad_enabled() or not conv.weight.requires_grad):
            if bn.training and bn.track_running_stats and not getattr(bn, "_nbt_external", False):
                bn.num_batches_tracked.add_(1)
            conv_w = conv.weight.view(3, 3)
            if x.requires_grad:
                raise RuntimeError("sample_down: the fused kernel does not produce a gradient for the image")
            return spf.sample_down(x, conv_w, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                   bn.training, self.size)
        w = conv.weight.view(conv.out_channels, conv.in_channels)
        x = torch.einsum("oc,bchw->bohw", w, x) + conv.bias.view(1, -1, 1, 1)
        x = F.relu(x)
        x = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)
        if bn.training and bn.track_running_stats:
            bn.num_batches_tracked.add_(1)
        return spf.resample_nearest(x, self.size)

    def forward_tokens(self, tokens, grid_hw):
        """tokens (B, gh*gw, Cin) -> stem on the token grid -> (B, gh, gw, Cout) channels-last."""
        B, T, E = tokens.shape
        conv, bn = self.stem[0], self.stem[2]
        rows = F.relu(F.linear(tokens.reshape(B * T, E), conv.weight.view(conv.out_channels, E), conv.bias))
        if bn.training and bn.track_running_stats and not getattr(bn, "_nbt_external", False):
            bn.num_batches_tracked.add_(1)
        rows = spf.batch_norm(rows, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps)
        return rows.view(B, grid_hw[0], grid_hw[1], conv.out_channels)


class Net2DBillinear(nn.Module):
    def __init__(self, num_classes: int, dual_head: bool, backbone_2d_kwargs=dict()):
        super().__init__()
        kw = backbone_2d_kwargs
        self.feat_channels = 96      # channels lifted onto the points
        self.hidden_channels = 768   # ViT width
        # the reference hard-codes (370, 1226) (image_models_billinear.py:74,77); other
        # lift sizes (384x1248, 900x1600) are throughput-only shapes
        ls = kw.get("lift_size", (370, 1226))
        self.lift_size = ls if isinstance(ls, str) else tuple(ls)

        self.sample_down = BilinearModule(in_features=3, out_features=3, interpolation_output_size=(384, 384))

        if kw.get("middle_feat_block_number", None) is not None:
            self.middle_feat_block_number = str(kw["middle_feat_block_number"])
        else:
            self.middle_feat_block_number = None
        if kw.get("late_feat_block_number", None) is not None:
            self.late_feat_block_number = str(kw["late_feat_block_number"])
        else:
            self.late_feat_block_number = None

        vit_kwargs = dict(remove_tokens_outputs=True)
        if kw.get("vit_depth", None) is not None:
            vit_kwargs["depth"] = int(kw["vit_depth"])
        if kw.get("skip_unused_blocks", True) and self.late_feat_block_number is not None:
            taps = [int(self.late_feat_block_number)] + ([int(self.middle_feat_block_number)] if self.middle_feat_block_number else [])
            vit_kwargs["last_block"] = max(taps)
        self.backbone = image_2d_distilled_transformer(pretrained=False, **vit_kwargs)
        if kw.get("IMAGE_PRETRAINED_PATH", "") != "":
            ckpt = torch.load(kw["IMAGE_PRETRAINED_PATH"], map_location="cpu", weights_only=True)["state_dict"]
            new_state_dict = OrderedDict((k.replace("backbone.", ""), v) for k, v in ckpt.items() if "backbone" in k)
            self.backbone.load_state_dict(new_state_dict)
        self.backbone.set_attention_impl(kw.get("attn_impl", "ftx"))
        if kw.get("vit_bf16", os.environ.get("FTX_VIT_BF16", "0") == "1"):
            self.backbone.set_bf16(True)   # BASELINE configs[4] "bf16 forward"; default is fp32 like the reference
        # Training on the GPU runs the trunk as HIP graphs, one per tapped segment (transformers.py): ~500 kernel launches
        # per step become 4 graph launches, which takes 7 ms off the host side of a step (batch 4: the host issue time
        # drops from ~24 to ~17 ms, so the step stays GPU-bound on a slow host; +26-30 % frames/s at batch 1-2).
        # vit_graphs=False or FTX_VIT_GRAPHS=0 turns it off.  Known limit (torch 2.10 / ROCm 7): capturing a module whose
        # parameters already went through an EAGER backward crashes in hipStreamEndCapture
        # (tools/probes/graph_recapture.py) -- so do not switch it on in the middle of a run.
        if kw.get("vit_graphs", os.environ.get("FTX_VIT_GRAPHS", "1") != "0") and self.late_feat_block_number is not None:
            self.backbone.graph_taps = sorted({int(self.late_feat_block_number)} | ({int(self.middle_feat_block_number)} if self.middle_feat_block_number else set()))
        # Parameters that can never receive a gradient (the reference needs
        # find_unused_parameters=True for them, TorchpackInterface.py:81): the final `norm`
        # (forward_blocks never applies it) and blocks past the last tap.  Their .grad stays None
        # in the reference, so its optimizer never touches them; freezing them is the same thing.
        for p in self.backbone.norm.parameters():
            p.requires_grad_(False)
        if self.backbone.last_block is not None:
            for i, blk in enumerate(self.backbone.blocks):
                if i > self.backbone.last_block:
                    for p in blk.parameters():
                        p.requires_grad_(False)

        self.up = nn.ModuleDict()
        if self.middle_feat_block_number:
            self.up[self.middle_feat_block_number] = BilinearModule(self.hidden_channels, self.feat_channels, self.lift_size)
        self.up[self.late_feat_block_number] = BilinearModule(self.hidden_channels, self.feat_channels, self.lift_size)

        if self.middle_feat_block_number and self.middle_feat_block_number != self.late_feat_block_number:
            # the middle tap's features reach the LiDAR branch detached (middle_fusion.py:102, early_fusion.py:105) and nothing
            # else consumes them, so this module's parameters never receive a gradient: frozen like the trunk's unused ones
            # (its BatchNorm running statistics still update in training forwards, as in the reference)
            for p in self.up[self.middle_feat_block_number].parameters():
                p.requires_grad_(False)

        self.linear = nn.Linear(self.feat_channels, num_classes)
        self.dual_head = dual_head
        if dual_head:
            self.linear2 = nn.Linear(self.feat_channels, num_classes)

    def _lift_hw(self, image_shape, lift_size=None):
        """Size of the (never materialised) up-sampled map.  The reference hard-codes (370, 1226)
        (image_models_billinear.py:74,77) and ignores `image_shape` (:101); `lift_size="image"` in the
        config or a per-batch `lift_size` in the data dict selects the batch's own image size instead, so one
        model serves KITTI- and NuScenes-shaped batches (BASELINE configs[4])."""
        if lift_size is None:
            lift_size = self.lift_size
        if isinstance(lift_size, str):
            if lift_size != "image":
                raise ValueError("lift_size must be (H, W) or 'image'")
            lift_size = tuple(int(v) for v in image_shape[-2:])
        h, w = int(lift_size[0]), int(lift_size[1])
        if h <= 0 or w <= 0:
            raise ValueError("lift_size must be positive")
        return h, w

    def get_img_feats(self, img_indices, block_id: str, image_shape: tuple, backbone_output: Dict, lift_ctx=None, lift_size=None):
        """reference image_models_billinear.py:88-126 -> (sum N, 96).

        `lift_ctx` is a dict that lives for ONE forward: the taps of that forward share the sort of the points by
        source cell (the lift's atomic-free backward) through it.  Nothing about a batch is kept on the module."""
        x = backbone_output[block_id]
        g = 384 // 16
        grid = self.up[block_id].forward_tokens(x, (g, g))
        idx, frame = pack_img_indices(img_indices, x.device)
        H, W = self._lift_hw(image_shape, lift_size)
        seg = None
        if torch.is_grad_enabled() and grid.requires_grad:
            key = ("seg", grid.shape[0], g, H, W)
            if lift_ctx is None:
                lift_ctx = {}
            # the cached entry holds the very tensors it was built from: `is` identity, never addresses
            hit = lift_ctx.get(key)
            if hit is None or hit[0] is not idx or hit[1] is not frame:
                hit = (idx, frame, spf.lift_segments(idx, frame, grid.shape[0], g, g, H, W))
                lift_ctx[key] = hit
            seg = hit[2]
        return spf.lift_gather(grid, idx, frame, H, W, seg)

    def forward(self, img, img_indices, on_middle=None, on_step=None, lift_size=None):
        """reference image_models_billinear.py:128-155.  `on_middle(feats)` is called with the lifted
        features of the middle tap as soon as that block has run (the LiDAR branch, on another
        stream, only waits for this and not for the rest of the ViT); `on_step()` after every issued
        chunk of work (the scheduler uses it to interleave the other branch's kernel launches)."""
        img_indices = pack_img_indices(img_indices, img.device)
        lift_ctx = {}     # per-forward: both taps share one sort of the points by source cell
        x = self.sample_down(img)
        if on_step is not None:
            on_step()
        middle = {}

        def tap(i, tokens):
            if self.middle_feat_block_number is not None and str(i) == self.middle_feat_block_number and self.middle_feat_block_number in self.up:
                middle["feats"] = self.get_img_feats(img_indices, self.middle_feat_block_number, img.shape, {self.middle_feat_block_number: tokens},
                                                        lift_ctx=lift_ctx, lift_size=lift_size)
                if on_middle is not None:
                    on_middle(middle["feats"])
            if on_step is not None:
                on_step()

        backbone_output = self.backbone.forward_blocks(x, on_block=tap)
        late_feats = self.get_img_feats(img_indices, self.late_feat_block_number, img.shape, backbone_output, lift_ctx=lift_ctx, lift_size=lift_size)
        x = spf.linear(late_feats, self.linear.weight, self.linear.bias)
        preds = {"img_feats": late_feats, "img_seg_logit": x}
        if self.dual_head:
            preds["img_seg_logit2"] = spf.linear(late_feats, self.linear2.weight, self.linear2.bias)
        if self.middle_feat_block_number:
            preds["img_middle_feats"] = middle["feats"]
        return preds
