"""SparseTensor / PointTensor containers and the per-batch coordinate manager.

torchsparse is not a dependency; these carry the attributes the reference's
model code touches (`.F`, `.C`, `.s`, `.coord_maps`, `.kernel_maps`,
`.check()`, `.cuda()`; PointTensor `.idx_query`, `.weights`,
`.additional_features`) -- see data/collate.py:67, models/utils.py:29-33,52-61,
88-96, models/spvcnn.py:193.

Data layout in HBM: voxel features are row-major (N, C) float32; coordinates
(N, 4) int32 [x, y, z, batch]; a kernel map is two dense neighbour tables
(K, N_out) and (K, N_in) int32 with -1 for "no neighbour", K-major so that a
wave's 32 output rows read 128 contiguous bytes per offset; the convolutions run
on its compacted pair list (pair_in / pair_out / koff, pos / pos_t).  Row order of
every level is ascending coordinate hash (= upstream's torch.unique order)."""
from __future__ import annotations

import numpy as np
import torch

from . import functional as Fn


class KernelMap:
    """Pair list of one (kernel size, input stride, stride) map: see csrc/ftx_spconv.hip."""
    __slots__ = ("nbr", "pos", "pos_t", "pair_in", "pair_out", "koff", "n_pairs", "n_in", "n_out", "out_coords", "kvol")

    def __init__(self, nbr, pos, pos_t, pair_in, pair_out, koff, n_pairs, n_in, n_out, out_coords):
        self.nbr, self.pos, self.pos_t, self.pair_in, self.pair_out, self.koff = nbr, pos, pos_t, pair_in, pair_out, koff
        self.n_pairs, self.n_in, self.n_out, self.out_coords = n_pairs, n_in, n_out, out_coords
        self.kvol = nbr.shape[0]


class CoordinateManager:
    """Hash tables and kernel maps of one batch, built once and shared by every
    layer of the level (the reference caches them on the tensors:
    models/utils.py:60-61 copies `kernel_maps`, torchsparse conv3d fills them)."""

    def __init__(self):
        self.tables = {}       # stride -> HashTable over sphash(coords at that stride)
        self.coords = {}       # stride -> (N,4) int32
        self.kernel_maps = {}  # (ks, cur_stride, stride) -> KernelMap
        self._offsets = {}

    def offsets(self, ks, stride, device):
        key = (ks, stride)
        if key not in self._offsets:
            self._offsets[key] = torch.from_numpy(Fn.kernel_offsets(ks, stride)).to(device)
        return self._offsets[key]

    def table(self, stride):
        if stride not in self.tables:
            self.tables[stride] = Fn.HashTable(Fn.sphash(self.coords[stride]))
        return self.tables[stride]

    def kernel_map(self, ks, cur_stride, stride) -> KernelMap:
        key = (ks, cur_stride, stride)
        km = self.kernel_maps.get(key)
        if km is not None:
            return km
        coords = self.coords[cur_stride]
        n_in = coords.shape[0]
        table = self.table(cur_stride)
        off = self.offsets(ks, cur_stride, coords.device)
        if stride == 1:
            out_coords = coords
        else:
            new_stride = cur_stride * stride
            if new_stride in self.coords:
                out_coords = self.coords[new_stride]
            else:
                down = Fn.downsample_coords(coords, new_stride)
                uniq, first, cnt = Fn.unique_sorted(Fn.sphash(down))
                n_out = int(cnt.item())  # the one host sync per level: sizes the next level's tensors
                out_coords = Fn.gather_coords(down, first[:n_out].contiguous())
                self.coords[new_stride] = out_coords
        nbr = Fn.kernel_map_build(out_coords, off, table)
        pos, koff = Fn.kernel_map_count(nbr)
        # a strided kernel-2 map joins every input voxel to exactly one (parent, offset): P = N_in;
        # for the submanifold maps the pair count is data dependent (one host sync)
        n_pairs = n_in if (ks == stride and ks == 2) else int(koff[-1].item())
        pos, pos_t, pair_in, pair_out = Fn.kernel_map_pairs(nbr, pos, n_in, n_pairs)
        km = KernelMap(nbr, pos, pos_t, pair_in, pair_out, koff, n_pairs, n_in, out_coords.shape[0], out_coords)
        self.kernel_maps[key] = km
        return km


class SparseTensor:
    def __init__(self, feats, coords, stride=1):
        self.F = feats
        self.C = coords
        self.s = stride
        self.coord_maps = {}
        self.kernel_maps = {}
        self.cm = None  # CoordinateManager, attached by initial_voxelize

    def check(self):
        if self.s not in self.coord_maps:
            self.coord_maps[self.s] = self.C

    def to(self, device):
        self.F = self.F.to(device, non_blocking=True)
        self.C = self.C.to(device, non_blocking=True)
        return self

    def cuda(self):
        return self.to("cuda")

    def derive(self, feats, coords=None, stride=None):
        """New tensor on the same coordinate system (shares the caches, like torchsparse)."""
        out = SparseTensor(feats, self.C if coords is None else coords, self.s if stride is None else stride)
        out.coord_maps, out.kernel_maps, out.cm = self.coord_maps, self.kernel_maps, self.cm
        return out


class PointTensor:
    def __init__(self, feats, coords, idx_query=None, weights=None):
        self.F = feats
        self.C = coords
        self.idx_query = idx_query if idx_query is not None else {}
        self.weights = weights if weights is not None else {}
        self.additional_features = {"idx_query": {}, "counts": {}}

    def to(self, device):
        self.F = self.F.to(device)
        self.C = self.C.to(device)
        return self

    def cuda(self):
        return self.to("cuda")


def cat(tensors):
    """torchsparse.cat: channel concat of voxel tensors on identical coordinates."""
    return tensors[0].derive(torch.cat([t.F for t in tensors], 1))
