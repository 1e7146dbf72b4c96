"""CPU oracle for the FusionTransformer per-frame fusion forward/backward.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under ``fusiontransformer_amd/`` does.

It restates, with plain numpy / torch-CPU ops, the algorithm the reference
runs for the hot path.  Citations are ``file:line`` under ``/root/reference``.
The reference's own model path cannot be imported in the build container
(torchsparse v1.1.0 and timm==0.4.9 are absent: ``docker/Dockerfile:33``,
``setup.py:13``), so the torchsparse / timm arithmetic is restated from their
published algorithms.

PARITY STATUS
  * pinned by reference code run in the build container (tests/golden/):
    nearest-resample index rule + BilinearModule + lift gather
    (``models/image_models_billinear.py:8-24,88-126``), weighted CE / KL loss
    mix (``modules/SemanticTrainer.py:158-178``), SegIoU (``models/metric.py``),
    voxel coordinates (``data/utils/augmentation_3d.py:4-53``), projection
    (``data/semantic_kitti/preprocess.py:93-126``).
  * PARITY UNPINNED: everything reached through torchsparse (hash, query,
    voxelize, devoxelize, trilinear weights, kernel maps, sparse conv) and the
    timm ViT block arithmetic.  The reference holds no tests, golden vectors or
    fixtures for them (``FusionTransformer/tests/test_dataset.py`` is an
    unrelated template).  These functions follow the upstream sources as
    documented per function below and are checked for internal consistency
    (dense-conv equivalence, fp64 gradcheck) in tests/test_oracle.py.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------
# 1. coordinate hashing / query  (torchsparse v1.1.0 spf.sphash, sphashquery)
#    call sites: models/utils.py:19-21,44-50,74-80
# --------------------------------------------------------------------------
_FNV_OFFSET = np.uint64(14695981039346656037)
_FNV_PRIME = np.uint64(1099511628211)
_MASK60 = np.uint64(0x0FFFFFFFFFFFFFFF)


def _fnv(cols):
    """FNV-1a over 4 int32 words (zero-extended), folded to 60 bits."""
    with np.errstate(over="ignore"):
        h = np.full(cols[0].shape, _FNV_OFFSET, dtype=np.uint64)
        for c in cols:
            h = h ^ c.astype(np.int32).view(np.uint32).astype(np.uint64)
            h = h * _FNV_PRIME
        h = (h >> np.uint64(60)) ^ (h & _MASK60)
    return h.view(np.int64)


def sphash(coords, offsets=None):
    """coords (N,4) int [x,y,z,b] -> (N,) int64; with offsets (K,3) -> (K,N).

    models/utils.py:19 (plain) and :74-78 (kernel-offset form)."""
    c = np.ascontiguousarray(np.asarray(coords), dtype=np.int32)
    if offsets is None:
        return _fnv([c[:, 0], c[:, 1], c[:, 2], c[:, 3]])
    off = np.asarray(offsets, dtype=np.int32)
    out = np.empty((off.shape[0], c.shape[0]), dtype=np.int64)
    for k in range(off.shape[0]):
        out[k] = _fnv([c[:, 0] + off[k, 0], c[:, 1] + off[k, 1], c[:, 2] + off[k, 2], c[:, 3]])
    return out


def sphashquery(hash_query, hash_target):
    """Index of each query hash in hash_target, -1 when absent (models/utils.py:21,50,80)."""
    q = np.asarray(hash_query)
    t = np.asarray(hash_target)
    if t.size == 0:
        return np.full(q.shape, -1, dtype=np.int64)
    order = np.argsort(t, kind="stable")
    ts = t[order]
    pos = np.searchsorted(ts, q.reshape(-1))
    pos_c = np.minimum(pos, ts.size - 1)
    hit = ts[pos_c] == q.reshape(-1)
    out = np.where(hit, order[pos_c], -1).astype(np.int64)
    return out.reshape(q.shape)


def spcount(idx, n):
    """Histogram of voxel index per point, ignoring -1 (models/utils.py:22,51)."""
    idx = np.asarray(idx)
    return np.bincount(idx[idx >= 0], minlength=n).astype(np.int32)


def kernel_offsets(kernel_size, tensor_stride=1):
    """torchsparse KernelRegion(kernel_size, tensor_stride, 1).get_kernel_offset().

    odd kernel: x fastest; even kernel: z fastest (models/utils.py:71-72 uses
    KernelRegion(2, s, 1); every spnn.Conv3d builds one internally)."""
    single = (np.arange(-kernel_size // 2 + 1, kernel_size // 2 + 1) * tensor_stride).tolist()
    if kernel_size % 2 == 1:
        offs = [[x, y, z] for z in single for y in single for x in single]
    else:
        offs = [[x, y, z] for x in single for y in single for z in single]
    return np.array(offs, dtype=np.int32)


# --------------------------------------------------------------------------
# 2. voxelize / devoxelize / trilinear weights (differentiable torch-CPU)
# --------------------------------------------------------------------------
def spvoxelize(feats, idx, counts):
    """Scatter-mean: out[idx[i]] += feats[i] / counts[idx[i]] (models/utils.py:24-27,58)."""
    idx_t = torch.as_tensor(np.asarray(idx), dtype=torch.long)
    cnt_t = torch.as_tensor(np.asarray(counts), dtype=feats.dtype)
    valid = idx_t >= 0
    idx_v = idx_t[valid]
    contrib = feats[valid] / cnt_t[idx_v].unsqueeze(1)
    out = torch.zeros(cnt_t.shape[0], feats.shape[1], dtype=feats.dtype)
    return out.index_add(0, idx_v, contrib)


def calc_ti_weights(pc, idx_query, scale=1):
    """Trilinear weights, (8,N) float32 (models/utils.py:81-82).

    Corner order follows kernel_offsets(2, scale) (z fastest).  float32 throughout, in the operation order of
    torchsparse v1.1.0 calc_ti_weights (it works in the dtype of `pc`): three-factor product, / scale**3, zero where
    the neighbour is absent, / (sum + 1e-8)."""
    pc = np.asarray(pc, dtype=np.float32)[:, :3]
    s32 = np.float32(scale)
    if scale != 1:
        pc_floor = np.floor(pc / s32) * s32
    else:
        pc_floor = np.floor(pc)
    pc_ceil = pc_floor + s32
    lo = pc - pc_floor   # weight towards the +offset corner
    hi = pc_ceil - pc    # weight towards the 0 corner
    ws = []
    for bx in (0, 1):
        for by in (0, 1):
            for bz in (0, 1):
                ws.append(((lo[:, 0] if bx else hi[:, 0]) * (lo[:, 1] if by else hi[:, 1])) * (lo[:, 2] if bz else hi[:, 2]))
    w = np.stack(ws, 0).astype(np.float32)
    if scale != 1:
        w = w / (s32 * s32 * s32)
    w[np.asarray(idx_query) == -1] = 0
    tot = np.zeros(w.shape[1], dtype=np.float32)
    for c in range(8):
        tot = tot + w[c]
    w = w / (tot + np.float32(1e-8))
    return w.astype(np.float32)


def spdevoxelize(feats, idx, weights):
    """out[i] = sum_k w[i,k] * feats[idx[i,k]] (idx<0 skipped) (models/utils.py:87,99)."""
    idx_t = torch.as_tensor(np.asarray(idx), dtype=torch.long)
    w_t = torch.as_tensor(np.asarray(weights), dtype=feats.dtype)
    out = torch.zeros(idx_t.shape[0], feats.shape[1], dtype=feats.dtype)
    for k in range(idx_t.shape[1]):
        m = idx_t[:, k] >= 0
        g = torch.zeros_like(out)
        g[m] = feats[idx_t[m, k]]
        out = out + w_t[:, k : k + 1] * g
    return out


# --------------------------------------------------------------------------
# 3. containers (torchsparse SparseTensor / PointTensor as used by the path)
# --------------------------------------------------------------------------
class SparseTensor:
    def __init__(self, feats, coords, stride=1):
        self.F = feats
        self.C = np.asarray(coords)
        self.s = stride
        self.coord_maps = {}
        self.kernel_maps = {}

    def check(self):
        if self.s not in self.coord_maps:
            self.coord_maps[self.s] = self.C


class PointTensor:
    def __init__(self, feats, coords, idx_query=None, weights=None):
        self.F = feats
        self.C = np.asarray(coords)
        self.idx_query = idx_query if idx_query is not None else {}
        self.weights = weights if weights is not None else {}
        self.additional_features = {"idx_query": {}, "counts": {}}


# --------------------------------------------------------------------------
# 4. point <-> voxel ops  (models/utils.py:15-106)
# --------------------------------------------------------------------------
def initial_voxelize(z, init_res, after_res):
    """models/utils.py:15-35."""
    new_float_coord = np.concatenate([(z.C[:, :3] * init_res) / after_res, z.C[:, -1:]], 1).astype(np.float32)
    pc_hash = sphash(np.floor(new_float_coord).astype(np.int32))
    sparse_hash = np.unique(pc_hash)
    idx_query = sphashquery(pc_hash, sparse_hash)
    counts = spcount(idx_query, len(sparse_hash))
    inserted_coords = spvoxelize(torch.from_numpy(np.floor(new_float_coord)), idx_query, counts)
    inserted_coords = torch.round(inserted_coords).int().numpy()
    inserted_feat = spvoxelize(z.F, idx_query, counts)
    new_tensor = SparseTensor(inserted_feat, inserted_coords, 1)
    new_tensor.check()
    z.additional_features["idx_query"][1] = idx_query
    z.additional_features["counts"][1] = counts
    z.C = new_float_coord
    return new_tensor


def _floor_to_stride(zc, s):
    c3 = np.floor(zc[:, :3] / np.float32(s)).astype(np.int32) * s
    return np.concatenate([c3, zc[:, -1:].astype(np.int32)], 1)


def point_to_voxel(x, z):
    """models/utils.py:40-63."""
    if z.additional_features["idx_query"].get(x.s) is None:
        pc_hash = sphash(_floor_to_stride(z.C, x.s))
        sparse_hash = sphash(x.C)
        idx_query = sphashquery(pc_hash, sparse_hash)
        counts = spcount(idx_query, x.C.shape[0])
        z.additional_features["idx_query"][x.s] = idx_query
        z.additional_features["counts"][x.s] = counts
    else:
        idx_query = z.additional_features["idx_query"][x.s]
        counts = z.additional_features["counts"][x.s]
    inserted_feat = spvoxelize(z.F, idx_query, counts)
    new_tensor = SparseTensor(inserted_feat, x.C, x.s)
    new_tensor.coord_maps = x.coord_maps
    new_tensor.kernel_maps = x.kernel_maps
    return new_tensor


def voxel_to_point(x, z, nearest=False):
    """models/utils.py:68-106."""
    if z.idx_query.get(x.s) is None or z.weights.get(x.s) is None:
        off = kernel_offsets(2, x.s)
        old_hash = sphash(_floor_to_stride(z.C, x.s), off)
        pc_hash = sphash(x.C)
        idx_query = sphashquery(old_hash, pc_hash)  # (8,N)
        weights = np.ascontiguousarray(calc_ti_weights(z.C, idx_query, scale=x.s).T)
        idx_query = np.ascontiguousarray(idx_query.T)
        if nearest:
            weights[:, 1:] = 0.0
            idx_query[:, 1:] = -1
        new_feat = spdevoxelize(x.F, idx_query, weights)
        new_tensor = PointTensor(new_feat, z.C, idx_query=z.idx_query, weights=z.weights)
        new_tensor.additional_features = z.additional_features
        new_tensor.idx_query[x.s] = idx_query
        new_tensor.weights[x.s] = weights
        z.idx_query[x.s] = idx_query
        z.weights[x.s] = weights
    else:
        new_feat = spdevoxelize(x.F, z.idx_query.get(x.s), z.weights.get(x.s))
        new_tensor = PointTensor(new_feat, z.C, idx_query=z.idx_query, weights=z.weights)
        new_tensor.additional_features = z.additional_features
    return new_tensor


# --------------------------------------------------------------------------
# 5. sparse convolution (torchsparse v1.1.0 spnn.Conv3d / functional conv3d)
#    call sites: models/spvcnn.py:26-30,42-46,57-72,99-101
# --------------------------------------------------------------------------
def spdownsample(coords, ratio):
    """Stride-`ratio` output coordinates, ordered by ascending hash.

    floor(c/ratio)*ratio on xyz, hash, torch.unique (sorted), mean of the
    (identical) member coordinates, round (SURVEY 2.2 'stride-2 coordinate
    downsample'; upstream torchsparse/nn/functional/downsample.py)."""
    c = np.asarray(coords, dtype=np.int32)
    new = np.concatenate([(np.floor(c[:, :3].astype(np.float32) / ratio) * ratio).astype(np.int32), c[:, 3:]], 1)
    h = sphash(new)
    _, first = np.unique(h, return_index=True)
    return new[first]


def build_kernel_map(coords_in, cur_stride, kernel_size, stride):
    """Returns (idx_query (K,N_out) of input rows or -1, out_coords).

    stride==1: out_coords = coords_in.  stride>1: out_coords = spdownsample."""
    off = kernel_offsets(kernel_size, cur_stride)
    out_coords = coords_in if stride == 1 else spdownsample(coords_in, stride * cur_stride)
    hash_query = sphash(out_coords, off)
    hash_target = sphash(coords_in)
    idx_query = sphashquery(hash_query, hash_target)
    return idx_query, out_coords


def sparseconv_op(feats, kernel, idx_query, n_out, transpose):
    """Gather - matmul - scatter-add over the kernel offsets.

    forward:   out[o] += feats[idx_query[k,o]] @ kernel[k]
    transpose: out[idx_query[k,o]] += feats[o] @ kernel[k]   (n_out = rows of the
               finer tensor; the map is the paired down-conv's)."""
    out = torch.zeros(n_out, kernel.shape[-1], dtype=feats.dtype)
    for k in range(idx_query.shape[0]):
        m = idx_query[k] >= 0
        if not m.any():
            continue
        in_rows = torch.from_numpy(idx_query[k][m])
        out_rows = torch.from_numpy(np.nonzero(m)[0])
        if transpose:
            in_rows, out_rows = out_rows, in_rows
        out = out.index_add(0, out_rows, feats[in_rows] @ kernel[k])
    return out


class Conv3d(nn.Module):
    """spnn.Conv3d: no bias, weight (K^3, inc, outc) ((inc,outc) for k=1)."""

    def __init__(self, inc, outc, kernel_size=3, stride=1, dilation=1, transpose=False):
        super().__init__()
        self.in_channels, self.out_channels = inc, outc
        self.kernel_size, self.stride, self.dilation, self.t = kernel_size, stride, dilation, transpose
        self.k = kernel_size ** 3
        self.kernel = nn.Parameter(torch.zeros(self.k, inc, outc)) if self.k > 1 else nn.Parameter(torch.zeros(inc, outc))
        std = 1.0 / math.sqrt(outc if transpose else inc * self.k)
        self.kernel.data.uniform_(-std, std)

    def forward(self, x):
        ks, s = self.kernel_size, self.stride
        if ks == 1 and s == 1:
            out = SparseTensor(x.F @ self.kernel, x.C, x.s)
            out.coord_maps, out.kernel_maps = x.coord_maps, x.kernel_maps
            out.check()
            return out
        if not self.t:
            key = "k%s_os%d_s%d_d%d" % (ks, x.s, s, self.dilation)
            km = x.kernel_maps.get(key)
            if km is None:
                idx_query, out_coords = build_kernel_map(x.C, x.s, ks, s)
                km = (idx_query, out_coords)
                x.kernel_maps[key] = km
            idx_query, out_coords = km
            out = SparseTensor(sparseconv_op(x.F, self.kernel, idx_query, out_coords.shape[0], False), out_coords, x.s * s)
            out.coord_maps, out.kernel_maps = x.coord_maps, x.kernel_maps
            out.check()
            return out
        original_stride = x.s // s
        key = "k%s_os%d_s%d_d%d" % (ks, original_stride, s, self.dilation)
        idx_query, _ = x.kernel_maps[key]
        fine_coords = x.coord_maps[original_stride]
        out = SparseTensor(sparseconv_op(x.F, self.kernel, idx_query, fine_coords.shape[0], True), fine_coords, original_stride)
        out.coord_maps, out.kernel_maps = x.coord_maps, x.kernel_maps
        out.check()
        return out


class BatchNorm(nn.BatchNorm1d):
    """spnn.BatchNorm = BatchNorm1d over the voxel rows."""

    def forward(self, x):
        out = SparseTensor(super().forward(x.F), x.C, x.s)
        out.coord_maps, out.kernel_maps = x.coord_maps, x.kernel_maps
        return out


class ReLU(nn.ReLU):
    def forward(self, x):
        out = SparseTensor(F.relu(x.F), x.C, x.s)
        out.coord_maps, out.kernel_maps = x.coord_maps, x.kernel_maps
        return out


def sparse_cat(tensors):
    """torchsparse.cat (models/middle_fusion.py:53,57,65,69)."""
    out = SparseTensor(torch.cat([t.F for t in tensors], 1), tensors[0].C, tensors[0].s)
    out.coord_maps, out.kernel_maps = tensors[0].coord_maps, tensors[0].kernel_maps
    return out


# --------------------------------------------------------------------------
# 6. SPVCNN (models/spvcnn.py:22-233)
# --------------------------------------------------------------------------
class BasicConvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(Conv3d(inc, outc, ks, stride, dilation), BatchNorm(outc), ReLU(True))

    def forward(self, x):
        return self.net(x)


class BasicDeconvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1):
        super().__init__()
        self.net = nn.Sequential(Conv3d(inc, outc, ks, stride, transpose=True), BatchNorm(outc), ReLU(True))

    def forward(self, x):
        return self.net(x)


class ResidualBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(
            Conv3d(inc, outc, ks, stride, dilation), BatchNorm(outc), ReLU(True),
            Conv3d(outc, outc, ks, 1, dilation), BatchNorm(outc))
        self.downsample = nn.Sequential() if (inc == outc and stride == 1) else nn.Sequential(
            Conv3d(inc, outc, 1, stride, 1), BatchNorm(outc))
        self.relu = ReLU(True)

    def forward(self, x):
        a, b = self.net(x), self.downsample(x)
        s = SparseTensor(a.F + b.F, a.C, a.s)
        s.coord_maps, s.kernel_maps = a.coord_maps, a.kernel_maps
        return self.relu(s)


class SPVCNN(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        cr = kwargs.get("cr", 1.0)
        cs = [int(cr * x) for x in [32, 32, 64, 128, 256, 256, 128, 96, 96]]
        self.cs = cs
        if "pres" in kwargs and "vres" in kwargs:
            self.pres, self.vres = kwargs["pres"], kwargs["vres"]
        else:
            self.pres = self.vres = 1
        self.stem = nn.Sequential(
            Conv3d(4, cs[0], 3, 1), BatchNorm(cs[0]), ReLU(True),
            Conv3d(cs[0], cs[0], 3, 1), BatchNorm(cs[0]), ReLU(True))

        def stage(i, o):
            return nn.Sequential(BasicConvolutionBlock(i, i, 2, 2, 1), ResidualBlock(i, o, 3, 1, 1), ResidualBlock(o, o, 3, 1, 1))

        self.stage1, self.stage2 = stage(cs[0], cs[1]), stage(cs[1], cs[2])
        self.stage3, self.stage4 = stage(cs[2], cs[3]), stage(cs[3], cs[4])

        def up(i, o, skip):
            return nn.ModuleList([BasicDeconvolutionBlock(i, o, 2, 2),
                                  nn.Sequential(ResidualBlock(o + skip, o, 3, 1, 1), ResidualBlock(o, o, 3, 1, 1))])

        self.up1, self.up2 = up(cs[4], cs[5], cs[3]), up(cs[5], cs[6], cs[2])
        self.up3, self.up4 = up(cs[6], cs[7], cs[1]), up(cs[7], cs[8], cs[0])
        self.point_transforms = nn.ModuleList([
            nn.Sequential(nn.Linear(cs[0], cs[4]), nn.BatchNorm1d(cs[4]), nn.ReLU(True)),
            nn.Sequential(nn.Linear(cs[4], cs[6]), nn.BatchNorm1d(cs[6]), nn.ReLU(True)),
            nn.Sequential(nn.Linear(cs[6], cs[8]), nn.BatchNorm1d(cs[8]), nn.ReLU(True))])
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.dropout = nn.Dropout(0.3, True)
        self.dropout_masks = None  # optional injected masks {'y1': (N4,C), 'y3': (N2,C)} for parity runs

    def _drop(self, feats, name):
        if self.dropout_masks is not None and self.training:
            return feats * self.dropout_masks[name] / (1.0 - 0.3)
        return self.dropout(feats)

    def backbone(self, x, fuse_early=None, fuse_middle=None):
        """models/spvcnn.py:191-233; fusion adds at early_fusion.py:39 / middle_fusion.py:48."""
        z = PointTensor(x.F, x.C.astype(np.float32))
        x0 = initial_voxelize(z, self.pres, self.vres)
        x0 = self.stem(x0)
        z0 = voxel_to_point(x0, z, nearest=False)
        if fuse_early is not None:
            z0.F = z0.F + fuse_early
        x1 = point_to_voxel(x0, z0)
        x1 = self.stage1(x1)
        x2 = self.stage2(x1)
        x3 = self.stage3(x2)
        x4 = self.stage4(x3)
        z1 = voxel_to_point(x4, z0)
        z1.F = z1.F + self.point_transforms[0](z0.F)
        if fuse_middle is not None:
            z1.F = z1.F + fuse_middle
        y1 = point_to_voxel(x4, z1)
        y1.F = self._drop(y1.F, "y1")
        y1 = self.up1[0](y1)
        y1 = sparse_cat([y1, x3])
        y1 = self.up1[1](y1)
        y2 = self.up2[0](y1)
        y2 = sparse_cat([y2, x2])
        y2 = self.up2[1](y2)
        z2 = voxel_to_point(y2, z1)
        z2.F = z2.F + self.point_transforms[1](z1.F)
        y3 = point_to_voxel(y2, z2)
        y3.F = self._drop(y3.F, "y3")
        y3 = self.up3[0](y3)
        y3 = sparse_cat([y3, x1])
        y3 = self.up3[1](y3)
        y4 = self.up4[0](y3)
        y4 = sparse_cat([y4, x0])
        y4 = self.up4[1](y4)
        z3 = voxel_to_point(y4, z2)
        z3.F = z3.F + self.point_transforms[2](z2.F)
        self.last_index = dict(x0=x0, x1=x1, x2=x2, x3=x3, x4=x4, z=z)
        return z3.F

    def forward(self, x):
        return self.backbone(x)


# --------------------------------------------------------------------------
# 7. image branch: BilinearModule, DeiT blocks, lift gather
#    models/image_models_billinear.py:8-155, models/transformers.py:16-45
# --------------------------------------------------------------------------
def nearest_src_index(n_out, n_in):
    """nn.Upsample(size) (= nearest) source index, float32 scale."""
    scale = np.float32(n_in) / np.float32(n_out)
    dst = np.arange(n_out, dtype=np.float32)
    return np.minimum(np.floor(dst * scale).astype(np.int64), n_in - 1)


class BilinearModule(nn.Module):
    """Conv1x1 -> ReLU -> BN2d -> nearest resample (image_models_billinear.py:8-24)."""

    def __init__(self, in_features, out_features, interpolation_output_size):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(in_features, out_features, kernel_size=1), nn.ReLU(True), nn.BatchNorm2d(out_features))
        self.size = tuple(interpolation_output_size)

    def forward(self, x):
        x = self.stem(x)
        rows = torch.from_numpy(nearest_src_index(self.size[0], x.shape[2]))
        cols = torch.from_numpy(nearest_src_index(self.size[1], x.shape[3]))
        return x[:, :, rows][:, :, :, cols]


class Attention(nn.Module):
    """timm 0.4.9 vision_transformer.Attention."""

    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class Image2DTransformer(nn.Module):
    """DeiT-base distilled p16/384 trunk (models/transformers.py:11-45,90-100)."""

    def __init__(self, img_size=384, patch_size=16, embed_dim=768, depth=12, num_heads=12, remove_tokens_outputs=True):
        super().__init__()
        self.remove_tokens_outputs = remove_tokens_outputs
        self.patch_embed = PatchEmbed(img_size, patch_size, 3, embed_dim)
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 2, embed_dim))
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)  # present in the state_dict, unused by forward_blocks
        for p in (self.cls_token, self.dist_token, self.pos_embed):
            nn.init.trunc_normal_(p, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward_blocks(self, x):
        x = self.patch_embed(x)
        B = x.shape[0]
        x = torch.cat((self.cls_token.expand(B, -1, -1), self.dist_token.expand(B, -1, -1), x), dim=1)
        x = x + self.pos_embed
        outputs = {}
        for i, block in enumerate(self.blocks):
            x = block(x)
            outputs[str(i)] = x[:, 2:, :] if self.remove_tokens_outputs else x
        return outputs


class Net2DBillinear(nn.Module):
    """models/image_models_billinear.py:26-155."""

    def __init__(self, num_classes, dual_head, backbone_2d_kwargs=None):
        super().__init__()
        kw = dict(backbone_2d_kwargs or {})
        self.feat_channels, self.hidden_channels = 96, 768
        self.lift_size = tuple(kw.get("lift_size", (370, 1226)))
        self.sample_down = BilinearModule(3, 3, (384, 384))
        depth = int(kw.get("vit_depth", 12))
        self.backbone = Image2DTransformer(depth=depth)
        mid = kw.get("middle_feat_block_number", None)
        late = kw.get("late_feat_block_number", None)
        self.middle_feat_block_number = str(mid) if mid is not None else None
        self.late_feat_block_number = str(late) if late is not None else None
        self.up = nn.ModuleDict()
        if self.middle_feat_block_number:  # NB: the string "0" is truthy, as in the reference (:69)
            self.up[self.middle_feat_block_number] = BilinearModule(768, 96, self.lift_size)
        self.up[self.late_feat_block_number] = BilinearModule(768, 96, self.lift_size)
        self.linear = nn.Linear(96, num_classes)
        self.dual_head = dual_head
        if dual_head:
            self.linear2 = nn.Linear(96, num_classes)

    def get_img_feats(self, img_indices, block_id, backbone_output):
        x = backbone_output[block_id]
        B, N, E = x.shape
        x = x.transpose(1, 2).reshape(B, E, 384 // 16, 384 // 16)
        x = self.up[block_id](x)
        feats = []
        for i in range(B):
            idx = torch.as_tensor(np.asarray(img_indices[i]), dtype=torch.long)
            feats.append(x.permute(0, 2, 3, 1)[i][idx[:, 0], idx[:, 1]])
        return torch.cat(feats, 0)

    def forward(self, img, img_indices):
        x = self.sample_down(img)
        out = self.backbone.forward_blocks(x)
        late = self.get_img_feats(img_indices, self.late_feat_block_number, out)
        preds = {"img_feats": late, "img_seg_logit": self.linear(late)}
        if self.dual_head:
            preds["img_seg_logit2"] = self.linear2(late)
        if self.middle_feat_block_number:
            preds["img_middle_feats"] = self.get_img_feats(img_indices, self.middle_feat_block_number, out)
        return preds


# --------------------------------------------------------------------------
# 8. fusion models (models/{early,middle,late}_fusion.py) and build_model
# --------------------------------------------------------------------------
class Net3DSegFused(SPVCNN):
    """Net3DSeg of middle_fusion.py:10-88 / early_fusion.py:9-87."""

    def __init__(self, num_classes, dual_head, mode, backbone_3d_kwargs=None):
        super().__init__(**(backbone_3d_kwargs or {}))
        self.mode = mode
        if mode == "middle":
            self.middle_fusion_transform = nn.Sequential(nn.Linear(96, self.cs[4]), nn.BatchNorm1d(self.cs[4]), nn.ReLU(True))
        else:
            self.early_fusion_transform = nn.Sequential(nn.Linear(96, 32), nn.BatchNorm1d(32), nn.ReLU(True))
        self.linear = nn.Linear(self.cs[-1], num_classes)
        self.dual_head = dual_head
        if dual_head:
            self.linear2 = nn.Linear(self.cs[-1], num_classes)

    def forward(self, x, img_feats):
        if self.mode == "middle":
            feats = self.backbone(x, fuse_middle=self.middle_fusion_transform(img_feats))
        else:
            feats = self.backbone(x, fuse_early=self.early_fusion_transform(img_feats))
        preds = {"lidar_feats": feats, "lidar_seg_logit": self.linear(feats)}
        if self.dual_head:
            preds["lidar_seg_logit2"] = self.linear2(feats)
        return preds


class Net3DSegLate(nn.Module):
    """late_fusion.py:5-35."""

    def __init__(self, num_classes, dual_head, backbone_3d_kwargs=None):
        super().__init__()
        self.backbone = SPVCNN(**(backbone_3d_kwargs or {}))
        self.linear = nn.Linear(self.backbone.cs[-1], num_classes)
        self.dual_head = dual_head
        if dual_head:
            self.linear2 = nn.Linear(self.backbone.cs[-1], num_classes)

    def forward(self, x):
        feats = self.backbone(x)
        preds = {"lidar_feats": feats, "lidar_seg_logit": self.linear(feats)}
        if self.dual_head:
            preds["lidar_seg_logit2"] = self.linear2(feats)
        return preds


class FusionTransformer(nn.Module):
    """Early/Middle/LateFusionTransformer (middle_fusion.py:90-112 etc.)."""

    def __init__(self, mode, num_class, dual_head, backbone_3d_kwargs, backbone_2d_kwargs):
        super().__init__()
        self.mode, self.dual_head = mode, dual_head
        if mode == "late":
            self.lidar_backbone = Net3DSegLate(num_class, dual_head, backbone_3d_kwargs)
        else:
            self.lidar_backbone = Net3DSegFused(num_class, dual_head, mode, backbone_3d_kwargs)
        self.image_backbone = Net2DBillinear(num_class, dual_head, backbone_2d_kwargs)

    def forward(self, data_dict):
        pi = self.image_backbone(data_dict["img"], data_dict["img_indices"])
        if self.mode == "late":
            pl = self.lidar_backbone(data_dict["lidar"])
        else:
            pl = self.lidar_backbone(data_dict["lidar"], pi["img_middle_feats"].detach())
        out = {"lidar_seg_logit": pl["lidar_seg_logit"], "img_seg_logit": pi["img_seg_logit"]}
        if self.dual_head:
            out.update({"lidar_seg_logit2": pl["lidar_seg_logit2"], "img_seg_logit2": pi["img_seg_logit2"]})
        return out


def build_model(model_cfg):
    """models/build.py:68-88 for the fusion types; model_cfg = cfg.MODEL as a dict."""
    mode = {"LateFusionTransformer": "late", "MiddleFusionTransformer": "middle", "EarlyFusionTransformer": "early"}[model_cfg["TYPE"]]
    return FusionTransformer(mode, model_cfg["NUM_CLASSES"], model_cfg["DUAL_HEAD"], model_cfg, model_cfg)


# --------------------------------------------------------------------------
# 9. losses (modules/SemanticTrainer.py:158-178) and SegIoU (models/metric.py:37-68)
# --------------------------------------------------------------------------
def fusion_losses(preds, seg_label, class_weights, lambda_xm, dual_head, mix="additive"):
    """mix="additive": modules/SemanticTrainer.py:158-178; mix="torchpack": modules/SemanticTorchpackTrainer.py:70-106."""
    loss_3d = F.cross_entropy(preds["lidar_seg_logit"], seg_label.long(), weight=class_weights)
    loss_2d = F.cross_entropy(preds["img_seg_logit"], seg_label.long(), weight=class_weights)
    if lambda_xm > 0:
        l2 = preds["img_seg_logit2"] if dual_head else preds["img_seg_logit"]
        l3 = preds["lidar_seg_logit2"] if dual_head else preds["lidar_seg_logit"]
        xm2 = F.kl_div(F.log_softmax(l2, dim=1), F.softmax(preds["lidar_seg_logit"].detach(), dim=1), reduction="none").sum(1).mean()
        xm3 = F.kl_div(F.log_softmax(l3, dim=1), F.softmax(preds["img_seg_logit"].detach(), dim=1), reduction="none").sum(1).mean()
        if mix == "torchpack":
            loss_2d = (1 - lambda_xm) * loss_2d + lambda_xm * xm2
            loss_3d = (1 - lambda_xm) * loss_3d + lambda_xm * xm3
        else:
            loss_2d = loss_2d + lambda_xm * xm2
            loss_3d = loss_3d + lambda_xm * xm3
    return loss_2d, loss_3d


def confusion_matrix(logits, labels, num_classes, ignore_index=0):
    pred = logits.argmax(1)
    m = labels != ignore_index
    inds = num_classes * labels[m].long() + pred[m]
    return torch.bincount(inds, minlength=num_classes ** 2).reshape(num_classes, num_classes)


def iou_from_matrix(mat):
    h = mat.float()
    return torch.diag(h) / (h.sum(1) + h.sum(0) - torch.diag(h))


# ---------------------------------------------------------------------------------------------
# evaluation scatter-back
# ---------------------------------------------------------------------------------------------
def validate_batch(lidar_logit, img_logit, inverse_maps, orig_seg_labels, n_vox, class_labels):
    """Per-batch body of data/utils/validate.py:62-120 with USE_FUSION, numpy.

    argmax / softmax-sum ensemble per model point (validate.py:62-72), `x[inverse_map]` per frame
    (map_sparse_to_org, validate.py:10-11,93-103), label inverse map (validate.py:105-113,
    semantic_kitti_dataloader.py:92), and Evaluator.update (data/utils/evaluate.py:12-26): gt id 0 becomes
    num_classes, then sklearn's confusion_matrix(gt, pred, labels=class_labels) -- entries whose gt or pred id is
    not in `labels` are dropped, rows are gt, columns pred.
    Returns (pred_3d, pred_2d, pred_ens) per original point in original ids and the three matrices."""
    l3 = np.asarray(lidar_logit, dtype=np.float32)
    l2 = np.asarray(img_logit, dtype=np.float32)
    class_labels = np.asarray(class_labels)
    c = len(class_labels)
    v3, v2 = l3.argmax(1), l2.argmax(1)
    def softmax(x):
        e = np.exp(x - x.max(1, keepdims=True), dtype=np.float32)
        return e / e.sum(1, keepdims=True, dtype=np.float32)
    ve = (softmax(l2) + softmax(l3)).argmax(1)
    index_of = {int(lab): i for i, lab in reversed(list(enumerate(class_labels)))}
    mats = [np.zeros((c, c)) for _ in range(3)]
    outs = [[], [], []]
    left = 0
    for b, (inv, gt) in enumerate(zip(inverse_maps, orig_seg_labels)):
        right = left + int(n_vox[b])
        gt_o = class_labels[np.asarray(gt)].copy()
        gt_o[gt_o == 0] = c
        for k, v in enumerate((v3, v2, ve)):
            pred_o = class_labels[v[left:right][np.asarray(inv)]]
            outs[k].append(pred_o)
            for g, p in zip(gt_o, pred_o):
                if int(g) in index_of and int(p) in index_of:
                    mats[k][index_of[int(g)], index_of[int(p)]] += 1
        left = right
    return [np.concatenate(o) for o in outs], mats


# ---------------------------------------------------------------- dataset-side 3-D augmentation
def augment_and_scale_3d_np(points, scale, full_scale, rot=None, transl_u=None):
    """data/utils/augmentation_3d.py:22-53 given the random draws (rot: the (3,3) float32 matrix after noise / flips / z rotation,
    transl_u: the rand(3) of the translation).  Pinned by tests/golden/voxel_coords_augmented.npz (the reference function run here).

    The rounding of `points.dot(rot_matrix)` (:41) is stated explicitly instead of being left to whatever BLAS is linked: one fused
    multiply-add per step of K = 3, t = x*r0j; t = fma(y, r1j, t); t = fma(z, r2j, t) -- fma(a, b, c) evaluated as
    float32(float64(a) * float64(b) + float64(c)), exact up to a double rounding that needs a 29-bit coincidence."""
    points = np.asarray(points, dtype=np.float32)
    if rot is not None:
        rot = np.asarray(rot, dtype=np.float32)
        x, y, z = (points[:, i].astype(np.float64) for i in range(3))
        cols = []
        for j in range(3):
            t = (x * np.float64(rot[0, j])).astype(np.float32)
            t = (y * np.float64(rot[1, j]) + t.astype(np.float64)).astype(np.float32)
            t = (z * np.float64(rot[2, j]) + t.astype(np.float64)).astype(np.float32)
            cols.append(t)
        points = np.stack(cols, 1)
    coords = points * np.float32(scale)
    coords = coords - coords.min(0)
    if transl_u is not None:
        room = np.clip(np.float32(full_scale) - coords.max(0) - np.float32(0.001), np.float32(0), None)      # float32 (:50)
        offset = room.astype(np.float64) * np.asarray(transl_u, dtype=np.float64)
        coords = (coords.astype(np.float64) + offset).astype(np.float32)                                     # float32 += float64 (:51)
    return coords
