"""SPVCNN sparse-voxel U-Net on libftx.

Mirror of the reference's FusionTransformer/models/spvcnn.py:22-233: same
class names, constructor arguments, attribute names (hence the same
state_dict keys: `stem.0.kernel`, `stage1.1.net.1.running_mean`, ...) and the
same wiring.  `spnn.Conv3d / BatchNorm / ReLU` are the classes below; at run
time each Conv->BN(->+residual)->ReLU chain is executed as one sparse-conv
launch plus one fused BN/residual/ReLU pass instead of four separate ops."""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as spf
from ..sparse import PendingIndex, PointTensor, PreparedIndex, SparseTensor, cat, drain, index_stream as _index_stream
from .utils import initial_voxelize, initial_voxelize_steps, point_index, point_to_voxel, voxel_index, voxel_to_point

__all__ = ["SPVCNN", "Conv3d", "BatchNorm", "ReLU"]

_FUSE_CONV_BN = os.environ.get("FTX_FUSE_CONV_BN", "1") != "0"


class Conv3d(nn.Module):
    """spnn.Conv3d (torchsparse v1.1.0): no bias, kernel (K^3, inc, outc), (inc, outc) when K=1."""

    def __init__(self, inc, outc, kernel_size=3, stride=1, dilation=1, bias=False, transpose=False):
        super().__init__()
        if bias:
            raise NotImplementedError("the reference never enables the conv bias")
        if dilation != 1:
            raise NotImplementedError("dilation != 1 is not used by the reference (spvcnn.py passes 1 everywhere)")
        self.in_channels, self.out_channels = inc, outc
        self.kernel_size, self.stride, self.dilation, self.t = kernel_size, stride, dilation, transpose
        self.k = kernel_size ** 3
        self.kernel = nn.Parameter(torch.zeros(self.k, inc, outc)) if self.k > 1 else nn.Parameter(torch.zeros(inc, outc))
        self.init_weight()

    def init_weight(self):
        std = 1.0 / math.sqrt(self.out_channels if self.t else self.in_channels * self.k)
        self.kernel.data.uniform_(-std, std)

    def forward(self, x: SparseTensor) -> SparseTensor:
        ks, s = self.kernel_size, self.stride
        if ks == 1 and s == 1:
            out = x.derive(spf.rows_matmul(x.F, self.kernel))
            out.check()
            return out
        if not self.t:
            km = x.cm.kernel_map(ks, x.s, s)
            feats = spf.sparse_conv(x.F, self.kernel, km, False)
            out = x.derive(feats, km.out_coords, x.s * s)
        else:
            original_stride = x.s // s
            km = x.cm.kernel_maps.get((ks, original_stride, s))
            if km is None:
                raise RuntimeError("transposed Conv3d needs the kernel map of the paired strided Conv3d")
            feats = spf.sparse_conv(x.F, self.kernel, km, True)
            out = x.derive(feats, x.cm.coords[original_stride], original_stride)
        out.check()
        return out


class BatchNorm(nn.BatchNorm1d):
    """spnn.BatchNorm: BatchNorm1d over the voxel rows; `fused` adds the residual and ReLU in the same pass."""

    def fused(self, feats, residual=None, relu=False):
        if self.training and self.track_running_stats and self.num_batches_tracked is not None and not getattr(self, "_nbt_external", False):
            self.num_batches_tracked.add_(1)
        return spf.batch_norm(feats, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                              self.momentum, self.eps, residual=residual, relu=relu)

    def forward(self, x):
        if isinstance(x, SparseTensor):
            return x.derive(self.fused(x.F))
        return self.fused(x)


class ReLU(nn.ReLU):
    """spnn.ReLU."""

    def forward(self, x):
        if isinstance(x, SparseTensor):
            return x.derive(F.relu(x.F))
        return F.relu(x)


def _conv_bn(conv, bn, x, residual=None, relu=True):
    """Conv3d -> BatchNorm (-> + residual) (-> ReLU).  In training a k>1 convolution and its BatchNorm run as one autograd node whose
    reduce pass also produces the batch statistics (functional.conv_bn_train); FTX_FUSE_CONV_BN=0 keeps the two separate nodes."""
    ks, s = conv.kernel_size, conv.stride
    if bn.training and _FUSE_CONV_BN and not (ks == 1 and s == 1) and x.F.is_cuda:
        if not conv.t:
            km = x.cm.kernel_map(ks, x.s, s)
            coords, stride = km.out_coords, x.s * s
        else:
            original_stride = x.s // s
            km = x.cm.kernel_maps.get((ks, original_stride, s))
            if km is None:
                raise RuntimeError("transposed Conv3d needs the kernel map of the paired strided Conv3d")
            coords, stride = x.cm.coords[original_stride], original_stride
        if bn.track_running_stats and bn.num_batches_tracked is not None and not getattr(bn, "_nbt_external", False):
            bn.num_batches_tracked.add_(1)
        feats = spf.conv_bn_train(x.F, conv.kernel, km, conv.t, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                  residual=residual, relu=relu)
        out = x.derive(feats, coords, stride)
        out.check()
        return out
    y = conv(x)
    return y.derive(bn.fused(y.F, residual=residual, relu=relu))


def _linear_bn_relu(seq, feats):
    """nn.Sequential(Linear, BatchNorm1d, ReLU) on point rows (spvcnn.py:164-180)."""
    return seq[1].fused(spf.linear(feats, seq[0].weight, seq[0].bias), relu=True)


class BasicConvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride), BatchNorm(outc), ReLU(True))

    def forward(self, x):
        return _conv_bn(self.net[0], self.net[1], x, relu=True)


class BasicDeconvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1):
        super().__init__()
        self.net = nn.Sequential(Conv3d(inc, outc, kernel_size=ks, stride=stride, transpose=True), BatchNorm(outc), ReLU(True))

    def forward(self, x):
        return _conv_bn(self.net[0], self.net[1], x, relu=True)


class ResidualBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(
            Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride), BatchNorm(outc), ReLU(True),
            Conv3d(outc, outc, kernel_size=ks, dilation=dilation, stride=1), BatchNorm(outc))
        self.downsample = nn.Sequential() if (inc == outc and stride == 1) else nn.Sequential(
            Conv3d(inc, outc, kernel_size=1, dilation=1, stride=stride), BatchNorm(outc))
        self.relu = ReLU(True)

    def forward(self, x):
        # relu(net(x) + downsample(x)), spvcnn.py:77-79; the add and ReLU ride on the last BN pass
        if len(self.downsample) == 0:
            shortcut = x.F
        else:
            shortcut = _conv_bn(self.downsample[0], self.downsample[1], x, relu=False).F
        h = _conv_bn(self.net[0], self.net[1], x, relu=True)
        return _conv_bn(self.net[3], self.net[4], h, residual=shortcut, relu=True)


class SPVCNN(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        cr = kwargs.get("cr", 1.0)
        cs = [32, 32, 64, 128, 256, 256, 128, 96, 96]
        cs = [int(cr * x) for x in cs]
        self.cs = cs
        if "pres" in kwargs and "vres" in kwargs:
            self.pres = kwargs["pres"]
            self.vres = kwargs["vres"]
        else:
            self.pres = 1
            self.vres = self.pres

        self.stem = nn.Sequential(
            Conv3d(4, cs[0], kernel_size=3, stride=1), BatchNorm(cs[0]), ReLU(True),
            Conv3d(cs[0], cs[0], kernel_size=3, stride=1), BatchNorm(cs[0]), ReLU(True))

        def stage(i, o):
            return nn.Sequential(BasicConvolutionBlock(i, i, ks=2, stride=2, dilation=1),
                                 ResidualBlock(i, o, ks=3, stride=1, dilation=1),
                                 ResidualBlock(o, o, ks=3, stride=1, dilation=1))

        self.stage1 = stage(cs[0], cs[1])
        self.stage2 = stage(cs[1], cs[2])
        self.stage3 = stage(cs[2], cs[3])
        self.stage4 = stage(cs[3], cs[4])

        def up(i, o, skip):
            return nn.ModuleList([BasicDeconvolutionBlock(i, o, ks=2, stride=2),
                                  nn.Sequential(ResidualBlock(o + skip, o, ks=3, stride=1, dilation=1),
                                                ResidualBlock(o, o, ks=3, stride=1, dilation=1))])

        self.up1 = up(cs[4], cs[5], cs[3])
        self.up2 = up(cs[5], cs[6], cs[2])
        self.up3 = up(cs[6], cs[7], cs[1])
        self.up4 = up(cs[7], cs[8], cs[0])

        self.point_transforms = nn.ModuleList([
            nn.Sequential(nn.Linear(cs[0], cs[4]), BatchNorm(cs[4]), nn.ReLU(True)),
            nn.Sequential(nn.Linear(cs[4], cs[6]), BatchNorm(cs[6]), nn.ReLU(True)),
            nn.Sequential(nn.Linear(cs[6], cs[8]), BatchNorm(cs[8]), nn.ReLU(True)),
        ])
        self.weight_initialization()
        self.dropout = nn.Dropout(0.3, True)
        # optional injected keep-masks {'y1': (N4,C), 'y3': (N2,C)} so a train-mode run can be
        # compared with the oracle (Dropout RNG streams differ between CPU and GPU)
        self.dropout_masks = None

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _drop(self, feats, name):
        if self.dropout_masks is not None and self.training:
            return feats * self.dropout_masks[name] / (1.0 - 0.3)
        return self.dropout(feats)

    def _stem(self, x):
        x = _conv_bn(self.stem[0], self.stem[1], x, relu=True)
        return _conv_bn(self.stem[3], self.stem[4], x, relu=True)

    def _backbone(self, x, fuse_early=None, fuse_middle=None):
        """spvcnn.py:191-233 with the fusion adds of early_fusion.py:39 / middle_fusion.py:48."""
        steps = self._backbone_steps(x, fuse_early, fuse_middle)
        while True:
            try:
                next(steps)
            except StopIteration as done:
                return done.value

    def _backbone_steps(self, x, fuse_early=None, fuse_middle=None):
        """The same forward as a generator that yields at stage boundaries, so a scheduler can interleave
        the issue of this branch with the image branch (see _fusion_common.run_fusion).  It yields
        "need_early" / "need_middle" right before it touches the image features."""
        prepared = getattr(x, "prepared", None)
        if isinstance(prepared, PendingIndex):
            # started ahead of this forward (prepare(wait=False)) and parked at a host read: finish it here, on its own stream
            while prepared.done is None:
                yield "sync"
                prepared.step()
            prepared = prepared.done
        if prepared is not None:
            # built ahead of this forward on the index stream (prepare()): wait for it, and tell the allocator this stream uses it
            z, x0 = prepared.take(torch.cuda.current_stream() if x.F.is_cuda else None)
            x.prepared = None
        else:
            z, x0 = yield from self._index_steps(x)
        yield "voxelized"
        x0 = self._stem(x0)
        z0 = voxel_to_point(x0, z, nearest=False)
        if fuse_early is not None:
            yield "need_early"
            z0.F = z0.F + (fuse_early() if callable(fuse_early) else fuse_early)
        yield "stem"

        x1 = point_to_voxel(x0, z0)
        x1 = self.stage1(x1)
        yield "stage1"
        x2 = self.stage2(x1)
        yield "stage2"
        x3 = self.stage3(x2)
        yield "stage3"
        x4 = self.stage4(x3)
        z1 = voxel_to_point(x4, z0)
        z1.F = z1.F + _linear_bn_relu(self.point_transforms[0], z0.F)
        if fuse_middle is not None:
            yield "need_middle"
            z1.F = z1.F + (fuse_middle() if callable(fuse_middle) else fuse_middle)
        yield "stage4"

        y1 = point_to_voxel(x4, z1)
        y1.F = self._drop(y1.F, "y1")
        y1 = self.up1[0](y1)
        y1 = cat([y1, x3])
        y1 = self.up1[1](y1)
        yield "up1"

        y2 = self.up2[0](y1)
        y2 = cat([y2, x2])
        y2 = self.up2[1](y2)
        z2 = voxel_to_point(y2, z1)
        z2.F = z2.F + _linear_bn_relu(self.point_transforms[1], z1.F)
        yield "up2"

        y3 = point_to_voxel(y2, z2)
        y3.F = self._drop(y3.F, "y3")
        y3 = self.up3[0](y3)
        y3 = cat([y3, x1])
        y3 = self.up3[1](y3)
        yield "up3"

        y4 = self.up4[0](y3)
        y4 = cat([y4, x0])
        y4 = self.up4[1](y4)
        yield "up4"
        z3 = voxel_to_point(y4, z2)
        z3.F = z3.F + _linear_bn_relu(self.point_transforms[2], z2.F)
        self.last_index = dict(x0=x0, x1=x1, x2=x2, x3=x3, x4=x4, z=z)
        return z3.F

    def _index_steps(self, x, ahead=False):
        """Everything of a batch that depends on its coordinates only: the voxelisation of the points, the voxel hash, the
        coordinates and kernel maps of the five U-Net levels and (ahead=True) the point <-> voxel index structures of the strides
        the network visits.  Each data-dependent size is read back after a "sync" yield (6 per batch), so a scheduler can issue
        image-branch work instead of waiting for it."""
        coords = x.C
        if coords.dtype != torch.float32:
            coords = coords.float()
        z = PointTensor(x.F, coords.contiguous())
        if os.environ.get("FTX_EAGER_INDEX_READS") == "1":   # A/B aid: block on every read as it comes
            x0 = initial_voxelize(z, self.pres, self.vres)
        else:
            levels = None if os.environ.get("FTX_LAZY_LEVELS") == "1" else (1, 2, 4, 8, 16)   # A/B aid: 1 = level l+1 from level l, six reads
            x0 = yield from initial_voxelize_steps(z, self.pres, self.vres, levels=levels)
            yield from x0.cm.unet_levels_steps((1, 2, 4, 8, 16))
        if ahead:
            cm, seg = x0.cm, torch.is_grad_enabled()
            # the strides of voxel_to_point / point_to_voxel in _backbone_steps: x0 / y4 (1), x4 (16), y2 (4)
            point_index(cm, 1, z, cm.coords[1].shape[0], with_segments=seg)
            for s_ in (16, 4):
                n = cm.coords[s_].shape[0]
                point_index(cm, s_, z, n, with_segments=seg)
                voxel_index(cm, s_, z, n)
        return z, x0

    def prepare(self, x, ready=None, wait=True):
        """Build the coordinate structures of batch `x` (a SparseTensor as the forward takes it) NOW, on a stream of their own, and
        hang them on `x`; the forward that later receives `x` starts at the first convolution.  Meant to be called for batch i+1
        while step i is in flight (trainer.TrainStep(next_batch=...)): inside the forward each host read of the build waits for
        everything queued before it -- the previous step's backward.
        `ready`: an event after which the batch's tensors are valid (default: all work queued on the current stream so far).
        `wait=False`: issue the build up to its FIRST host read only and return without blocking (the one-pass level build: the sort
        and the level sizes on their way to the host); the forward resumes it from there.  The caller's thread never waits for the
        GPU in this form, which is what the training loop wants while the backward is still executing."""
        if getattr(x, "prepared", None) is not None or not x.F.is_cuda:
            return x
        with torch.cuda.device(x.F.device):      # the current device is per thread (prepare may run on a helper thread)
            s_idx = _index_stream(x.F.device)
            if ready is not None:
                s_idx.wait_event(ready)          # the batch was valid at `ready`: do not queue behind what the caller issued since
            else:
                s_idx.wait_stream(torch.cuda.current_stream())
        pending = PendingIndex(self._index_steps(x, ahead=True), s_idx, x.F.device, self.training)   # training: also the backward's sorted segments
        finished = pending.step()
        while wait and not finished:
            finished = pending.step()
        x.prepared = pending.done if finished else pending
        return x

    def forward(self, x):
        return self._backbone(x)
