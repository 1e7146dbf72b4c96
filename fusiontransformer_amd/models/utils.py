"""initial_voxelize / point_to_voxel / voxel_to_point on the GPU.

Same names, arguments and caching behaviour as the reference's
FusionTransformer/models/utils.py:15-106; every `spf.*` call there maps to the
function of the same name in fusiontransformer_amd.functional (libftx)."""
from __future__ import annotations

import torch

from .. import functional as spf
from ..sparse import CoordinateManager, HostRead, PointTensor, SparseTensor, drain

__all__ = ["initial_voxelize", "initial_voxelize_steps", "point_to_voxel", "voxel_to_point", "voxel_index", "point_index"]


def initial_voxelize(z: PointTensor, init_res, after_res) -> SparseTensor:
    """reference models/utils.py:15-35."""
    return drain(initial_voxelize_steps(z, init_res, after_res))


def initial_voxelize_steps(z: PointTensor, init_res, after_res, levels=None):
    """initial_voxelize as a generator that yields "sync" before the voxel count is read back (sparse.HostRead).

    `levels`: the strides the network will visit, starting with 1 (SPVCNN: (1, 2, 4, 8, 16)).  Every level's voxel set is then taken
    from the points in ONE pass (functional.levels_unique) and the single host read returns all level sizes; the CoordinateManager
    keeps the per-level hashes / first occurrences and builds each level's coordinates and hash table from them without another read
    (the lazy form -- level l+1 from level l, one read per level -- remains for `levels=None`).  Same sets, order and coordinates."""
    if init_res == after_res:
        new_float_coord = z.C
    else:
        new_float_coord = torch.cat([(z.C[:, :3] * init_res) / after_res, z.C[:, -1].view(-1, 1)], 1).contiguous()
    floored = spf.floor_coords(new_float_coord, 1)           # torch.floor(...).int()
    pc_hash = spf.sphash(floored)
    level_data = None
    if levels is not None:
        levels = tuple(int(s) for s in levels)
        if not levels or levels[0] != 1:
            raise ValueError("initial_voxelize: `levels` must start with stride 1")
        uniq_all, first_all, level_off, sorted_keys, order = spf.levels_unique(floored, levels)
        pending = HostRead(level_off)
        yield "sync"
        offs = pending.values()
        # per level: (sorted unique hashes, first-occurrence point rows, level index, the level's sorted keys, its points sorted by voxel)
        level_data = {s: (uniq_all[offs[i]:offs[i + 1]], first_all[offs[i]:offs[i + 1]], i, sorted_keys[i], order[i]) for i, s in enumerate(levels)}
        sparse_hash, first1 = level_data[1][:2]
        n_vox = sparse_hash.shape[0]
    else:
        uniq, first, cnt = spf.unique_sorted(pc_hash)        # torch.unique(pc_hash)
        pending = HostRead(cnt)
        yield "sync"
        n_vox = pending.value()
        sparse_hash = uniq[:n_vox].contiguous()
        first1 = first[:n_vox].contiguous()
    table = spf.HashTable(sparse_hash)
    idx_query = table.query(pc_hash)                         # spf.sphashquery(pc_hash, sparse_hash)
    counts = spf.spcount(idx_query, n_vox)
    # round(mean of identical integer coordinates) == the coordinates of any member
    inserted_coords = spf.gather_coords(floored, first1)
    seg = _level_segments(level_data, 1) if level_data is not None else spf.voxelize_segments(idx_query, n_vox)
    inserted_feat = spf.spvoxelize(z.F, idx_query, counts, seg)

    new_tensor = SparseTensor(inserted_feat, inserted_coords, 1)
    new_tensor.cm = CoordinateManager()
    new_tensor.cm.coords[1] = inserted_coords
    new_tensor.cm.tables[1] = table                          # keys == sphash(inserted_coords), row order == table rows
    if level_data is not None:
        new_tensor.cm.points, new_tensor.cm.level_data = floored, level_data
    new_tensor.check()
    z.additional_features["idx_query"][1] = idx_query
    z.additional_features["counts"][1] = counts
    z.additional_features.setdefault("vox_seg", {})[1] = seg
    z.C = new_float_coord
    return new_tensor


def _level_segments(level_data, stride):
    """Points sorted by their voxel at `stride`, from the one-pass level sort (no second sort)."""
    hashes, _, level, skeys, order = level_data[stride]
    return spf.level_segments(skeys, order, hashes, level)


def voxel_index(cm: CoordinateManager, stride: int, z: PointTensor, n_vox: int):
    """The coordinate-only half of point_to_voxel at `stride` (hash query, counts, sorted segments), cached on `z`."""
    pc_hash = spf.sphash(spf.floor_coords(z.C, stride))
    idx_query = cm.table(stride).query(pc_hash)              # sphashquery(pc_hash, sphash(x.C))
    z.additional_features["idx_query"][stride] = idx_query
    z.additional_features["counts"][stride] = spf.spcount(idx_query, n_vox)
    ld = getattr(cm, "level_data", None)
    z.additional_features.setdefault("vox_seg", {})[stride] = _level_segments(ld, stride) if (ld and stride in ld) else spf.voxelize_segments(idx_query, n_vox)


def point_index(cm: CoordinateManager, stride: int, z: PointTensor, n_vox: int, nearest=False, with_segments=True):
    """The coordinate-only half of voxel_to_point at `stride` (8 corner rows, trilinear weights and -- for the backward -- the
    (point, corner) entries sorted by voxel), cached on `z`."""
    off = cm.offsets(2, stride, z.F.device)                  # KernelRegion(2, x.s, 1)
    floored = spf.floor_coords(z.C, stride)
    # sphash(floored, off) + sphashquery against sphash(x.C), fused; (8, N) -> (N, 8)
    idx_query = spf.kernel_map_build(floored, off, cm.table(stride)).transpose(0, 1).contiguous()
    weights = spf.calc_ti_weights(z.C, idx_query, scale=stride)
    if nearest:
        weights[:, 1:] = 0.0
        idx_query[:, 1:] = -1
    z.idx_query[stride] = idx_query
    z.weights[stride] = weights
    z.additional_features.setdefault("devox_seg", {})[stride] = spf.devoxelize_segments(idx_query, weights, n_vox) if with_segments else None


def point_to_voxel(x: SparseTensor, z: PointTensor) -> SparseTensor:
    """reference models/utils.py:40-63."""
    if z.additional_features is None or z.additional_features.get("idx_query") is None \
            or z.additional_features["idx_query"].get(x.s) is None:
        voxel_index(x.cm, x.s, z, x.C.shape[0])
    idx_query = z.additional_features["idx_query"][x.s]
    counts = z.additional_features["counts"][x.s]
    seg = z.additional_features.setdefault("vox_seg", {}).get(x.s)
    inserted_feat = spf.spvoxelize(z.F, idx_query, counts, seg)
    return x.derive(inserted_feat)


def voxel_to_point(x: SparseTensor, z: PointTensor, nearest=False) -> PointTensor:
    """reference models/utils.py:68-106."""
    if z.idx_query is None or z.weights is None or z.idx_query.get(x.s) is None or z.weights.get(x.s) is None:
        point_index(x.cm, x.s, z, x.F.shape[0], nearest=nearest, with_segments=x.F.requires_grad)
    segs = z.additional_features.setdefault("devox_seg", {})
    if segs.get(x.s) is None and x.F.requires_grad:
        segs[x.s] = spf.devoxelize_segments(z.idx_query.get(x.s), z.weights.get(x.s), x.F.shape[0])
    new_feat = spf.spdevoxelize(x.F, z.idx_query.get(x.s), z.weights.get(x.s), segs.get(x.s))
    new_tensor = PointTensor(new_feat, z.C, idx_query=z.idx_query, weights=z.weights)
    new_tensor.additional_features = z.additional_features
    return new_tensor
