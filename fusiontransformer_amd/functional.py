"""Host-side operator surface over libftx: the torchsparse `spf.*` functions the
reference calls (models/utils.py:19-27,44-58,71-99), the kernel-map builder and
sparse convolution hidden inside `spnn.Conv3d` (models/spvcnn.py:26-30), the
fused BatchNorm(+residual)(+ReLU), and the fused nearest-upsample + lift gather
of `Net2DBillinear.get_img_feats` (models/image_models_billinear.py:113-124).

Same names and argument meaning as the reference's imports; tensors live on
the GPU, index tensors are int32 (torchsparse returns int64 and immediately
`.int()`s them, utils.py:22,51)."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, req, stream

I32, I64, F32 = torch.int32, torch.int64, torch.float32


def _empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


# Kernel workspaces (BatchNorm / weight-gradient partials, sort and scan temporaries) live in ONE growing buffer per (device, stream):
# a workspace is only touched by the kernels of the call that receives it, and calls on one stream execute in order, so the next
# call may reuse the bytes.  This takes two allocator round trips and a size query through ctypes out of every such call (~150 per
# training step).  While a stream is being captured into a HIP graph the buffer must not be baked in (it can be re-grown later), so
# capture falls back to a fresh allocation from the graph's pool.
_SCRATCH = {}
_WS_BYTES = {}


def _ws_bytes(fn_name, *args):
    key = (fn_name,) + args
    v = _WS_BYTES.get(key)
    if v is None:
        v = _WS_BYTES[key] = int(getattr(_lib.load(), fn_name)(*args))
        if len(_WS_BYTES) > 4096:
            _WS_BYTES.clear()
    return v


def _carve(like, *nbytes):
    """Raw device pointers of consecutive 256-byte aligned regions of the stream's scratch buffer: the temporaries of ONE call
    (pair rows, partial statistics, an intermediate gradient, kernel workspaces) that never leave it."""
    if torch.cuda.is_current_stream_capturing():
        # Under capture `_scratch` hands out a fresh allocation from the graph's private pool; this function returns raw addresses and
        # drops the tensor, so a later allocation of the same capture could be given the same block while kernels recorded earlier
        # still use it.  The callers (the Conv3d + BatchNorm node) also rely on the library's per-stream ticket counters, which must
        # not be baked into a graph that replays beside eager launches.  Neither is supported: fail instead of corrupting data.
        raise RuntimeError("fusiontransformer_amd: the fused Conv3d + BatchNorm node cannot be captured into a HIP graph "
                           "(its temporaries are carved from the stream's scratch buffer); run the LiDAR branch eagerly")
    offs, total = [], 0
    for n in nbytes:
        offs.append(total)
        total += (int(n) + 255) & ~255
    base = _scratch(max(total, 256), like).data_ptr()
    return [base + o for o in offs]


def _scratch(nbytes, like):
    dev = like.device
    if torch.cuda.is_current_stream_capturing():
        return torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    key = (dev.index, stream())
    buf = _SCRATCH.get(key)
    if buf is None or buf.shape[0] < nbytes:
        buf = _SCRATCH[key] = torch.empty((max(int(nbytes * 1.25), 1 << 22),), dtype=torch.uint8, device=dev)
    return buf


# The statistics kernels (BatchNorm, the convolution's reduce-with-statistics pass) keep a few ticket counters per (device, stream)
# (include/ftx.h, ftx_stream_scratch_*).  The caller owns that buffer: it is a torch allocation attached to the stream at its first use.
_TICKETS = {}


def _stream_scratch(st=None):
    """Make sure the current stream has its ticket buffer attached (a dict lookup after the first call); returns the raw stream."""
    if st is None:
        st = stream()
    key = (torch.cuda.current_device(), st)
    if key not in _TICKETS and not torch.cuda.is_current_stream_capturing():
        L = _lib.load()
        n = int(L.ftx_stream_scratch_bytes())
        buf = torch.empty((n,), dtype=torch.uint8, device=torch.device("cuda", key[0]))
        check(L.ftx_stream_scratch_attach(st, buf.data_ptr(), n), "ftx_stream_scratch_attach")
        _TICKETS[key] = buf
    return st


def reset_stream_scratch():
    """Clear the ticket counters of the current stream (after a kernel of the library died mid-flight on it)."""
    check(_lib.load().ftx_stream_scratch_reset(stream()), "ftx_stream_scratch_reset")


def release_stream_scratch():
    """Detach (and free) the ticket buffers of every stream this process attached one to."""
    L = _lib.load()
    for (dev, st) in list(_TICKETS):
        with torch.cuda.device(dev):
            check(L.ftx_stream_scratch_release(st), "ftx_stream_scratch_release")
        del _TICKETS[(dev, st)]


# ---------------------------------------------------------------- integer side
def sphash(coords: torch.Tensor, offsets: torch.Tensor | None = None) -> torch.Tensor:
    """spf.sphash: (N,4) int32 -> (N,) int64, or with (K,3) offsets -> (K,N)."""
    L = _lib.load()
    req(coords, I32, "sphash coords", 2)
    if coords.shape[1] != 4:
        raise ValueError("sphash: coords must be (N,4)")
    n = coords.shape[0]
    if offsets is None:
        out = _empty((n,), I64, coords)
        check(L.ftx_hash(ptr(coords), n, ptr(out), stream()), "ftx_hash")
        return out
    req(offsets, I32, "sphash offsets", 2)
    k = offsets.shape[0]
    out = _empty((k, n), I64, coords)
    check(L.ftx_hash_kernel(ptr(coords), n, ptr(offsets), k, ptr(out), stream()), "ftx_hash_kernel")
    return out


def floor_coords(pc: torch.Tensor, stride: int) -> torch.Tensor:
    """cat([floor(pc[:, :3] / s).int() * s, pc[:, -1].int()]) (models/utils.py:44-48)."""
    L = _lib.load()
    req(pc, F32, "floor_coords pc", 2)
    if pc.shape[1] != 4:
        raise ValueError("floor_coords: pc must be (N,4)")
    out = _empty(pc.shape, I32, pc)
    check(L.ftx_floor_coords(ptr(pc), pc.shape[0], int(stride), ptr(out), stream()), "ftx_floor_coords")
    return out


class HashTable:
    """Open-addressing table (64-bit keys -> int32 row) resident in HBM."""

    def __init__(self, keys: torch.Tensor):
        L = _lib.load()
        req(keys, I64, "HashTable keys", 1)
        self.n = keys.shape[0]
        self.capacity = int(L.ftx_hashtable_capacity(self.n))
        self.keys = _empty((self.capacity,), I64, keys)
        self.vals = _empty((self.capacity,), I32, keys)
        check(L.ftx_hashtable_build(ptr(keys), self.n, ptr(self.keys), ptr(self.vals), self.capacity, stream()), "ftx_hashtable_build")

    def query(self, q: torch.Tensor) -> torch.Tensor:
        L = _lib.load()
        req(q, I64, "HashTable query")
        out = _empty(q.shape, I32, q)
        check(L.ftx_hashtable_query(ptr(q), q.numel(), ptr(self.keys), ptr(self.vals), self.capacity, ptr(out), stream()), "ftx_hashtable_query")
        return out


def sphashquery(hash_query: torch.Tensor, hash_target: torch.Tensor) -> torch.Tensor:
    """spf.sphashquery: row of each query hash in hash_target, -1 if absent (int32)."""
    return HashTable(hash_target).query(hash_query)


def spcount(idx: torch.Tensor, n: int) -> torch.Tensor:
    L = _lib.load()
    req(idx, I32, "spcount idx", 1)
    out = _empty((int(n),), I32, idx)
    check(L.ftx_count(ptr(idx), idx.shape[0], ptr(out), int(n), stream()), "ftx_count")
    return out


def unique_sorted(keys: torch.Tensor):
    """torch.unique(keys) (ascending) plus the row of the first occurrence of each.

    Returns (uniq (n,), first_index (n,), n_unique (1,) int32 ON DEVICE); rows
    past n_unique are unspecified.  No host sync here."""
    L = _lib.load()
    req(keys, I64, "unique_sorted keys", 1)
    n = keys.shape[0]
    uniq = _empty((n,), I64, keys)
    first = _empty((n,), I32, keys)
    cnt = torch.zeros((1,), dtype=I32, device=keys.device)
    ws_bytes = _ws_bytes("ftx_unique_workspace_bytes", n)
    ws = _scratch(ws_bytes, keys)
    check(L.ftx_unique_sorted(ptr(keys), n, ptr(uniq), ptr(first), ptr(cnt), ptr(ws), ws_bytes, stream()), "ftx_unique_sorted")
    return uniq, first, cnt


def levels_unique(points: torch.Tensor, strides):
    """Every U-Net level's voxel set from the floored point coordinates in ONE pass (ftx_levels_unique): for each stride s the sorted
    unique hashes of floor_div(p, s) * s and the point row of each one's first occurrence, back to back, plus `level_off`
    (len(strides) + 1,) int32 ON THE DEVICE -- where each level's run starts.  One sort, no host sync here."""
    L = _lib.load()
    req(points, I32, "levels_unique points", 2)
    if points.shape[1] != 4:
        raise ValueError("levels_unique: points must be (N,4)")
    st = np.ascontiguousarray(np.asarray(strides, dtype=np.int32))
    nl, n = int(st.shape[0]), points.shape[0]
    if not (1 <= nl <= 8) or (st < 1).any():
        raise ValueError("levels_unique: 1..8 strides, each >= 1")
    uniq = _empty((nl * n,), I64, points)
    first = _empty((nl * n,), I32, points)
    level_off = _empty((nl + 1,), I32, points)
    sorted_keys = _empty((nl, n), I64, points)        # row l: level l's (tag | hash) keys, ascending
    order = _empty((nl, n), I32, points)              # row l: the points sorted by their voxel of level l (stable)
    ws_bytes = _ws_bytes("ftx_levels_workspace_bytes", n, nl)
    ws = _scratch(ws_bytes, points)
    check(L.ftx_levels_unique(ptr(points), n, st.ctypes.data, nl, ptr(uniq), ptr(first), ptr(level_off), ptr(sorted_keys), ptr(order), ptr(ws), ws_bytes,
                              stream()), "ftx_levels_unique")
    return uniq, first, level_off, sorted_keys, order


def level_segments(sorted_keys_row: torch.Tensor, order_row: torch.Tensor, hashes: torch.Tensor, level: int) -> "Segments":
    """The sorted segments of spvoxelize at one level from levels_unique's sort (no second sort): == Segments(idx_query, n_vox)."""
    L = _lib.load()
    req(sorted_keys_row, I64, "level_segments sorted keys", 1)
    req(order_row, I32, "level_segments order", 1)
    req(hashes, I64, "level_segments hashes", 1)
    m = hashes.shape[0]
    seg_off = _empty((m + 1,), I32, hashes)
    check(L.ftx_level_segments(ptr(sorted_keys_row), sorted_keys_row.shape[0], ptr(hashes), m, int(level), ptr(seg_off), stream()), "ftx_level_segments")
    return Segments.from_parts(order_row, seg_off, m)


def level_coords(points: torch.Tensor, first_index: torch.Tensor, stride: int) -> torch.Tensor:
    """Coordinates of one level: floor_div(points[first_index], stride) * stride, rows in the level's (hash) order."""
    L = _lib.load()
    req(points, I32, "level_coords points", 2)
    req(first_index, I32, "level_coords first_index", 1)
    out = _empty((first_index.shape[0], 4), I32, points)
    check(L.ftx_level_coords(ptr(points), ptr(first_index), first_index.shape[0], int(stride), ptr(out), stream()), "ftx_level_coords")
    return out


def sorted_rank(sorted_keys: torch.Tensor, n_sorted: torch.Tensor, queries: torch.Tensor) -> torch.Tensor:
    """Position of every query in sorted_keys[:n_sorted] (ascending, unique), -1 when absent; n_sorted is a (1,) int32 DEVICE tensor,
    so `unique_sorted` + `sorted_rank` give numpy.unique(return_index, return_inverse) without a host read."""
    L = _lib.load()
    req(sorted_keys, I64, "sorted_rank sorted", 1)
    req(n_sorted, I32, "sorted_rank n_sorted", 1)
    req(queries, I64, "sorted_rank queries", 1)
    rank = _empty((queries.shape[0],), I32, queries)
    check(L.ftx_sorted_rank(ptr(sorted_keys), ptr(n_sorted), sorted_keys.shape[0], ptr(queries), queries.shape[0], ptr(rank), stream()), "ftx_sorted_rank")
    return rank


def rotate_points(points: torch.Tensor, rot) -> torch.Tensor:
    """points (N,3) float32 @ rot (3,3) float32 (host array) with the rounding of numpy's `points.dot(rot)` (libftx)."""
    import ctypes
    import numpy as np
    L = _lib.load()
    req(points, F32, "rotate_points points", 2)
    if points.shape[1] != 3:
        raise ValueError("rotate_points: points must be (N, 3)")
    r = np.ascontiguousarray(np.asarray(rot, dtype=np.float32).reshape(9))
    out = _empty(tuple(points.shape), F32, points)
    check(L.ftx_rotate_points(ptr(points), points.shape[0], r.ctypes.data_as(ctypes.c_void_p), ptr(out), stream()), "ftx_rotate_points")
    return out


def downsample_coords(coords: torch.Tensor, ratio: int) -> torch.Tensor:
    L = _lib.load()
    req(coords, I32, "downsample coords", 2)
    out = torch.empty_like(coords)
    check(L.ftx_downsample_coords(ptr(coords), coords.shape[0], int(ratio), ptr(out), stream()), "ftx_downsample_coords")
    return out


def gather_coords(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    L = _lib.load()
    req(src, I32, "gather_coords src", 2)
    req(index, I32, "gather_coords index", 1)
    out = _empty((index.shape[0], 4), I32, src)
    check(L.ftx_gather_coords(ptr(src), ptr(index), index.shape[0], ptr(out), stream()), "ftx_gather_coords")
    return out


def kernel_offsets(kernel_size: int, tensor_stride: int = 1) -> np.ndarray:
    """torchsparse KernelRegion(kernel_size, tensor_stride).get_kernel_offset():
    odd kernels enumerate x fastest, even kernels z fastest."""
    single = (np.arange(-kernel_size // 2 + 1, kernel_size // 2 + 1) * tensor_stride).tolist()
    if kernel_size % 2 == 1:
        offs = [[x, y, z] for z in single for y in single for x in single]
    else:
        offs = [[x, y, z] for x in single for y in single for z in single]
    return np.array(offs, dtype=np.int32)


def kernel_map_build(out_coords: torch.Tensor, offsets: torch.Tensor, table: HashTable) -> torch.Tensor:
    """nbr (K, N_out) int32: row of out_coords[o]+offsets[k] among the table's keys, or -1."""
    L = _lib.load()
    req(out_coords, I32, "kernel_map out_coords", 2)
    req(offsets, I32, "kernel_map offsets", 2)
    n_out, k = out_coords.shape[0], offsets.shape[0]
    nbr = _empty((k, n_out), I32, out_coords)
    check(L.ftx_kernel_map_build(ptr(out_coords), n_out, ptr(offsets), k, ptr(table.keys), ptr(table.vals), table.capacity, ptr(nbr), stream()),
          "ftx_kernel_map_build")
    return nbr


def kernel_map_count(nbr: torch.Tensor):
    """Step 1 of the pair list: returns (pos scratch (K,N_out), koff (K+1,) int32 on the device).
    koff[-1] is the number of pairs; the caller reads it back (one host sync) to size the arrays."""
    L = _lib.load()
    req(nbr, I32, "kernel_map_count nbr", 2)
    k, n_out = nbr.shape
    pos = torch.empty_like(nbr)
    koff = _empty((k + 1,), I32, nbr)
    ws_bytes = _ws_bytes("ftx_kernel_map_count_workspace_bytes", n_out, k)
    ws = _scratch(ws_bytes, nbr)
    check(L.ftx_kernel_map_count(ptr(nbr), n_out, k, ptr(pos), ptr(koff), ptr(ws), ws_bytes, stream()), "ftx_kernel_map_count")
    return pos, koff


def kernel_map_pairs(nbr: torch.Tensor, pos: torch.Tensor, n_in: int, n_pairs: int):
    """Step 2: (pos (K,N_out), pos_t (K,N_in), pair_in (P,), pair_out (P,))."""
    L = _lib.load()
    k, n_out = nbr.shape
    pos_t = _empty((k, int(n_in)), I32, nbr)
    pair_in = _empty((int(n_pairs),), I32, nbr)
    pair_out = _empty((int(n_pairs),), I32, nbr)
    check(L.ftx_kernel_map_pairs(ptr(nbr), n_out, int(n_in), k, ptr(pos), ptr(pos_t), ptr(pair_in), ptr(pair_out), int(n_pairs), stream()),
          "ftx_kernel_map_pairs")
    return pos, pos_t, pair_in, pair_out


def calc_ti_weights(pc: torch.Tensor, idx_query: torch.Tensor, scale: int = 1) -> torch.Tensor:
    """spf.calc_ti_weights, already in the point-major (N,8) layout of utils.py:82-83."""
    L = _lib.load()
    req(pc, F32, "calc_ti_weights pc", 2)
    req(idx_query, I32, "calc_ti_weights idx", 2)
    n = pc.shape[0]
    if pc.shape[1] != 4 or idx_query.shape != (n, 8):
        raise ValueError("calc_ti_weights: pc must be (N,4) and idx (N,8)")
    w = _empty((n, 8), F32, pc)
    check(L.ftx_trilinear_weights(ptr(pc), ptr(idx_query), n, int(scale), ptr(w), stream()), "ftx_trilinear_weights")
    return w


# ---------------------------------------------------------------- voxelize / devoxelize
class Segments:
    """Entries sorted by destination row (ftx_segment_build): `order`, `seg_off`, `m` rows."""
    __slots__ = ("order", "seg_off", "m")

    @classmethod
    def from_parts(cls, order, seg_off, m):
        self = cls.__new__(cls)
        self.order, self.seg_off, self.m = order, seg_off, int(m)
        return self

    def __init__(self, keys: torch.Tensor, m: int):
        L = _lib.load()
        keys = req(keys.contiguous().view(-1), I32, "segment keys", 1)
        n = keys.shape[0]
        self.m = int(m)
        self.order = _empty((n,), I32, keys)
        self.seg_off = _empty((self.m + 1,), I32, keys)
        ws_bytes = _ws_bytes("ftx_segment_workspace_bytes", n, self.m)
        ws = _scratch(ws_bytes, keys)
        check(L.ftx_segment_build(ptr(keys), n, self.m, ptr(self.order), ptr(self.seg_off), ptr(ws), ws_bytes, stream()), "ftx_segment_build")


def voxelize_segments(idx: torch.Tensor, m: int) -> Segments:
    """Points sorted by their voxel: makes the scatter-mean of spvoxelize a gather-reduce."""
    return Segments(idx, m)


def devoxelize_segments(idx: torch.Tensor, weights: torch.Tensor, m: int) -> Segments:
    """(point, corner) entries sorted by voxel, zero-weight corners dropped: makes the scatter-add of
    spdevoxelize's backward a gather-reduce."""
    keys = torch.where(weights != 0, idx, torch.full_like(idx, -1))
    return Segments(keys, m)


class _Voxelize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, idx, counts, seg):
        L = _lib.load()
        feats = req(feats.contiguous(), F32, "spvoxelize feats", 2)
        req(idx, I32, "spvoxelize idx", 1)
        req(counts, I32, "spvoxelize counts", 1)
        n, c = feats.shape
        if idx.shape[0] != n:
            raise ValueError("spvoxelize: idx length != rows of feats")
        m = counts.shape[0]
        out = _empty((m, c), F32, feats)
        if seg is not None and c % 4 == 0:
            if seg.m != m or seg.order.shape[0] != n:
                raise ValueError("spvoxelize: segments do not match idx / counts")
            check(L.ftx_voxelize_fwd_sorted(ptr(feats), ptr(seg.order), ptr(seg.seg_off), n, c, m, ptr(out), stream()), "ftx_voxelize_fwd_sorted")
        else:
            check(L.ftx_voxelize_fwd(ptr(feats), ptr(idx), ptr(counts), n, c, m, ptr(out), stream()), "ftx_voxelize_fwd")
        ctx.save_for_backward(idx, counts)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L = _lib.load()
        idx, counts = ctx.saved_tensors
        grad_out = req(grad_out.contiguous(), F32, "spvoxelize grad", 2)
        m, c = grad_out.shape
        gf = _empty((ctx.n, c), F32, grad_out)
        check(L.ftx_voxelize_bwd(ptr(grad_out), ptr(idx), ptr(counts), ctx.n, c, m, ptr(gf), stream()), "ftx_voxelize_bwd")
        return gf, None, None, None


def spvoxelize(feats, idx, counts, seg=None):
    """spf.spvoxelize (scatter-mean of point rows into voxel rows); `seg` = voxelize_segments(idx, m)
    selects the atomic-free sorted form."""
    return _Voxelize.apply(feats, idx, counts, seg)


class _Devoxelize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, idx, weights, seg):
        L = _lib.load()
        feats = req(feats.contiguous(), F32, "spdevoxelize feats", 2)
        req(idx, I32, "spdevoxelize idx", 2)
        req(weights, F32, "spdevoxelize weights", 2)
        m, c = feats.shape
        n = idx.shape[0]
        if idx.shape != (n, 8) or weights.shape != (n, 8):
            raise ValueError("spdevoxelize: idx and weights must be (N,8)")
        out = _empty((n, c), F32, feats)
        check(L.ftx_devoxelize_fwd(ptr(feats), ptr(idx), ptr(weights), n, c, m, ptr(out), stream()), "ftx_devoxelize_fwd")
        ctx.save_for_backward(idx, weights)
        ctx.m, ctx.seg = m, seg
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L = _lib.load()
        idx, weights = ctx.saved_tensors
        grad_out = req(grad_out.contiguous(), F32, "spdevoxelize grad", 2)
        n, c = grad_out.shape
        gf = _empty((ctx.m, c), F32, grad_out)
        seg = ctx.seg
        if seg is not None:
            if seg.m != ctx.m or seg.order.shape[0] != n * 8:
                raise ValueError("spdevoxelize: segments do not match idx")
            check(L.ftx_devoxelize_bwd_sorted(ptr(grad_out), ptr(weights), ptr(seg.order), ptr(seg.seg_off), n, c, ctx.m, ptr(gf), stream()),
                  "ftx_devoxelize_bwd_sorted")
        else:
            check(L.ftx_devoxelize_bwd(ptr(grad_out), ptr(idx), ptr(weights), n, c, ctx.m, ptr(gf), stream()), "ftx_devoxelize_bwd")
        return gf, None, None, None


def spdevoxelize(feats, idx, weights, seg=None):
    """spf.spdevoxelize (8-corner weighted gather of voxel rows onto points); `seg` =
    devoxelize_segments(idx, weights, m) makes the backward atomic-free."""
    return _Devoxelize.apply(feats, idx, weights, seg)


# ---------------------------------------------------------------- sparse convolution
# When set to a list, every sparse-conv launch appends (kind, start_event, end_event,
# shape dict): HIP events recorded on the launch stream, read back by bench.py after the step.
LAUNCH_LOG = None


def _log_launch(kind, meta, launch):
    if LAUNCH_LOG is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = launch()
    e1.record()
    LAUNCH_LOG.append((kind, e0, e1, meta))
    return out


def _spconv_apply(A, W, gather, pos, koff, n_pairs, n_rows_out, co, w_transposed):
    """reduce(pairs_gemm(A[gather] @ W[k]), pos) -> (n_rows_out, co)."""
    L = _lib.load()
    rows_a, ca = A.shape
    kvol = koff.shape[0] - 1
    tmp = _empty((n_pairs, co), F32, A)
    out = _empty((n_rows_out, co), F32, A)

    def launch_gemm():
        check(L.ftx_spconv_pairs_gemm(ptr(A), rows_a, ptr(gather), ptr(W), int(w_transposed), ptr(koff), n_pairs, ca, co, kvol, ptr(tmp), stream()),
              "ftx_spconv_pairs_gemm")

    def launch_reduce():
        check(L.ftx_spconv_reduce(ptr(tmp), ptr(pos), n_rows_out, co, kvol, ptr(out), stream()), "ftx_spconv_reduce")

    meta = dict(pairs=n_pairs, n_out=n_rows_out, ca=ca, co=co, kvol=kvol)
    _log_launch("spconv_pairs_gemm", meta, launch_gemm)
    _log_launch("spconv_reduce", meta, launch_reduce)
    return out


def _spconv_direct(A, W, gather, scatter, koff, n_pairs, n_rows_out, co, w_transposed):
    """out[scatter[p]] = A[gather[p]] @ W[k(p)] in ONE launch, for maps whose destination side is a bijection of the pair list."""
    L = _lib.load()
    rows_a, ca = A.shape
    kvol = koff.shape[0] - 1
    if n_pairs != n_rows_out:
        raise ValueError("direct sparse conv: the pair list must cover every destination row exactly once")
    out = _empty((n_rows_out, co), F32, A)

    def launch():
        check(L.ftx_spconv_pairs_gemm_scatter(ptr(A), rows_a, ptr(gather), ptr(scatter), ptr(W), int(w_transposed), ptr(koff), n_pairs, ca, co, kvol,
                                              ptr(out), n_rows_out, stream()), "ftx_spconv_pairs_gemm_scatter")

    _log_launch("spconv_pairs_gemm", dict(pairs=n_pairs, n_out=n_rows_out, ca=ca, co=co, kvol=kvol, direct=True), launch)
    return out


_OSTAT = os.environ.get("FTX_OSTAT", "1") != "0"      # A/B aid: 0 = every convolution on the pair-list kernels
_OSTAT_ALL = os.environ.get("FTX_OSTAT", "1") == "all"   # every layer the kernel supports, not only where it is faster
_OSTAT_OK = {}


_OSTAT_MAX_ROWS = 64 * 4096      # one block per 64 output rows, at most 4096 partial rows in the statistics hand-over


def ostat_supported(ca, co, kvol, w_transposed=False, rows=0):
    """Does the one-launch output-stationary kernel (ftx_spconv_ostat) take this layer?  (ca in {4, 32, 64}, co in {32, 64},
    at most 262 144 output rows; anything else runs on the pair-list kernels)"""
    key = (int(ca), int(co), int(kvol), bool(w_transposed))
    v = _OSTAT_OK.get(key)
    if v is None:
        v = _OSTAT_OK[key] = bool(_lib.load().ftx_spconv_ostat_supported(key[0], key[1], key[2], int(key[3])))
    return v and _OSTAT and rows <= _OSTAT_MAX_ROWS


def ostat_preferred(ca, co, kvol, w_transposed=False, rows=0):
    """Where the output-stationary kernel is FASTER than pair GEMM + reduce on MI355X (tools/bench_spconv.py, profiles/r03_spconv_layer_micro.txt):
    forward convolutions with c_in <= 32 and c_out = 32 -- the stem, the 32 -> 32 layers of levels 1 and 2, the strided 32 -> 32 layers.
    It is bound by instruction issue (~23 instructions per pair), so 64-channel layers and the transposed-W data-gradient form lose
    (csrc/ftx_spconv_ostat.hip); FTX_OSTAT=all routes everything the kernel supports through it (tests, measurements)."""
    if not ostat_supported(ca, co, kvol, w_transposed, rows):
        return False
    if _OSTAT_ALL:
        return True
    return (not w_transposed) and co == 32 and ca <= 32


def _spconv_ostat(A, W, nbr, n_rows_out, co, w_transposed, flip, part=0, nb=0, pairs=0):
    """out[o] = sum_k A[nbr[k, o]] @ W[k] in ONE launch (no pair rows, no reduce pass); `part`: also the BatchNorm statistics."""
    L = _lib.load()
    rows_a, ca = A.shape
    kvol = nbr.shape[0]
    if nbr.shape[1] != n_rows_out:
        raise ValueError("ostat sparse conv: the neighbour table does not match the output rows")
    out = _empty((n_rows_out, co), F32, A)
    st = _stream_scratch() if part else stream()

    def launch():
        check(L.ftx_spconv_ostat(ptr(A), rows_a, ptr(nbr), n_rows_out, ptr(W), int(w_transposed), int(flip), ca, co, kvol, ptr(out), part, nb, st),
              "ftx_spconv_ostat")

    _log_launch("spconv_ostat", dict(pairs=pairs, n_out=n_rows_out, ca=ca, co=co, kvol=kvol, direct=True), launch)
    return out


def _spconv_wgrad(A, idx_a, G, idx_g, koff, n_pairs):
    L = _lib.load()
    rows_a, ca = A.shape
    rows_g, cg = G.shape
    kvol = koff.shape[0] - 1
    dW = _empty((kvol, ca, cg), F32, A)
    ws_bytes = _ws_bytes("ftx_spconv_pairs_wgrad_workspace_bytes", n_pairs, ca, cg, kvol)
    ws = _scratch(ws_bytes, A)

    def launch():
        check(L.ftx_spconv_pairs_wgrad(ptr(A), rows_a, ptr(idx_a), ptr(G), rows_g, ptr(idx_g), ptr(koff), n_pairs, ca, cg, kvol, ptr(dW), ptr(ws),
                                       ws_bytes, stream()), "ftx_spconv_pairs_wgrad")

    _log_launch("spconv_pairs_wgrad", dict(pairs=n_pairs, n_out=rows_g, ca=ca, co=cg, kvol=kvol), launch)
    return dW


def _conv_shapes(feats, kernel, km, transposed):
    kvol, ca, co = kernel.shape
    n_in, n_out = (km.n_out, km.n_in) if transposed else (km.n_in, km.n_out)
    if feats.shape != (n_in, ca) or km.kvol != kvol:
        raise ValueError(f"conv3d: shape mismatch feats {tuple(feats.shape)} kernel {tuple(kernel.shape)} map ({km.kvol},{n_in}->{n_out})")
    return kvol, ca, co, n_in, n_out


def _conv_forward(feats, kernel, km, transposed):
    kvol, ca, co, n_in, n_out = _conv_shapes(feats, kernel, km, transposed)
    if transposed and km.fine_bijective:
        # every fine row is the destination of exactly one pair: the GEMM epilogue writes `out` itself
        return _spconv_direct(feats, kernel, km.pair_out, km.pair_in, km.koff, km.n_pairs, n_out, co, 0)
    if not transposed and n_out > 0 and km.n_pairs > 0 and ostat_preferred(ca, co, kvol, rows=n_out):
        return _spconv_ostat(feats, kernel, km.nbr, n_out, co, 0, 0, pairs=km.n_pairs)
    gather, pos = (km.pair_out, km.pos_t) if transposed else (km.pair_in, km.pos)
    return _spconv_apply(feats, kernel, gather, pos, km.koff, km.n_pairs, n_out, co, 0)


def _conv_backward(feats, kernel, km, transposed, grad_out, need_feats, need_kernel):
    kvol, ca, co = kernel.shape
    g_feats = g_kernel = None
    in_side, out_side = (km.pair_out, km.pair_in) if transposed else (km.pair_in, km.pair_out)
    if need_feats:
        if not transposed and km.fine_bijective:
            g_feats = _spconv_direct(grad_out, kernel, km.pair_out, km.pair_in, km.koff, km.n_pairs, feats.shape[0], ca, 1)
        elif not transposed and km.submanifold and km.n_pairs > 0 and ostat_preferred(co, ca, kvol, True, rows=feats.shape[0]):
            # symmetric map: the data gradient is the same output-stationary kernel on the same table, read mirrored
            g_feats = _spconv_ostat(grad_out, kernel, km.nbr, feats.shape[0], ca, 1, 1, pairs=km.n_pairs)
        else:
            pos_in = km.pos if transposed else km.pos_t
            g_feats = _spconv_apply(grad_out, kernel, out_side, pos_in, km.koff, km.n_pairs, feats.shape[0], ca, 1)
    if need_kernel:
        g_kernel = _spconv_wgrad(feats, in_side, grad_out, out_side, km.koff, km.n_pairs)
    return g_feats, g_kernel


class _SparseConv(torch.autograd.Function):
    """out[o] = sum over pairs (k, i->o) of feats[i] @ kernel[k].

    `km` is a KernelMap (pair list); `transposed` swaps the roles of its two sides: the
    transposed conv of models/spvcnn.py:42-46 reuses the paired strided conv's map."""

    @staticmethod
    def forward(ctx, feats, kernel, km, transposed):
        feats = req(feats.contiguous(), F32, "conv3d feats", 2)
        kernel = req(kernel.contiguous(), F32, "conv3d kernel", 3)
        out = _conv_forward(feats, kernel, km, transposed)
        ctx.save_for_backward(feats, kernel)
        ctx.km, ctx.transposed = km, transposed
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feats, kernel = ctx.saved_tensors
        grad_out = req(grad_out.contiguous(), F32, "conv3d grad", 2)
        g_feats, g_kernel = _conv_backward(feats, kernel, ctx.km, ctx.transposed, grad_out, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return g_feats, g_kernel, None, None


def sparse_conv(feats, kernel, km, transposed=False):
    return _SparseConv.apply(feats, kernel, km, transposed)


# ---------------------------------------------------------------- dense rows (skinny GEMMs)
def _rows_gemm(A, W, w_transposed, bias, co):
    L = _lib.load()
    n, ca = A.shape
    out = _empty((n, co), F32, A)
    check(L.ftx_rows_gemm(ptr(A), n, ptr(W), int(w_transposed), ptr(bias), ca, co, ptr(out), stream()), "ftx_rows_gemm")
    return out


def _rows_wgrad(A, G):
    """A (n, ca)^T @ G (n, cg) -> (ca, cg)."""
    L = _lib.load()
    n, ca = A.shape
    cg = G.shape[1]
    dW = _empty((1, ca, cg), F32, A)
    ws_bytes = _ws_bytes("ftx_spconv_pairs_wgrad_workspace_bytes", n, ca, cg, 1)
    ws = _scratch(ws_bytes, A)
    check(L.ftx_spconv_pairs_wgrad(ptr(A), n, 0, ptr(G), n, 0, 0, n, ca, cg, 1, ptr(dW), ptr(ws), ws_bytes, stream()), "ftx_spconv_pairs_wgrad(dense)")
    return dW[0]


def _rows_ok(*channels):
    return all(c >= 4 and c % 4 == 0 for c in channels)


class _RowsLinear(torch.autograd.Function):
    """F.linear on (N, C) rows: out = x @ weight^T + bias, weight (co, ca) as nn.Linear stores it."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = req(x.contiguous(), F32, "linear x", 2)
        weight = req(weight.contiguous(), F32, "linear weight", 2)
        co, ca = weight.shape
        if x.shape[1] != ca:
            raise ValueError("linear: shape mismatch")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _rows_gemm(x, weight, 1, bias, co)

    @staticmethod
    def backward(ctx, go):
        x, weight = ctx.saved_tensors
        go = req(go.contiguous(), F32, "linear grad", 2)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _rows_gemm(go, weight, 0, None, weight.shape[1])
        if ctx.needs_input_grad[1]:
            gw = _rows_wgrad(go, x)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = colsum(go) if go.shape[1] % 4 == 0 else go.sum(0)
        return gx, gw, gb


class _RowsMatmul(torch.autograd.Function):
    """x (N, ca) @ kernel (ca, co): the kernel_size = 1 spnn.Conv3d."""

    @staticmethod
    def forward(ctx, x, kernel):
        x = req(x.contiguous(), F32, "matmul x", 2)
        kernel = req(kernel.contiguous(), F32, "matmul kernel", 2)
        ctx.save_for_backward(x, kernel)
        return _rows_gemm(x, kernel, 0, None, kernel.shape[1])

    @staticmethod
    def backward(ctx, go):
        x, kernel = ctx.saved_tensors
        go = req(go.contiguous(), F32, "matmul grad", 2)
        gx = _rows_gemm(go, kernel, 1, None, kernel.shape[0]) if ctx.needs_input_grad[0] else None
        gk = _rows_wgrad(x, go) if ctx.needs_input_grad[1] else None
        return gx, gk


def linear(x, weight, bias=None):
    """nn.Linear on point / voxel rows.  Skinny shapes (tens of thousands of rows, <= 384 channels)
    run on libftx's tile kernel; anything else is a plain library GEMM."""
    if x.dim() == 2 and _rows_ok(x.shape[1], weight.shape[0]) and max(weight.shape) <= 512:
        return _RowsLinear.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


def rows_matmul(x, kernel):
    if x.dim() == 2 and _rows_ok(*kernel.shape) and max(kernel.shape) <= 512:
        return _RowsMatmul.apply(x, kernel)
    return torch.matmul(x, kernel)


# ---------------------------------------------------------------- BatchNorm (+residual)(+ReLU)
class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu):
        L = _lib.load()
        x = req(x.contiguous(), F32, "bn x", 2)
        n, c = x.shape
        if residual is not None:
            residual = req(residual.contiguous(), F32, "bn residual", 2)
            if residual.shape != x.shape:
                raise ValueError("bn: residual shape mismatch")
        for t, nm in ((gamma, "gamma"), (beta, "beta")):
            req(t, F32, "bn " + nm, 1)
            if t.shape[0] != c:
                raise ValueError("bn: parameter length != channels")
        y = torch.empty_like(x)
        mean = _empty((c,), F32, x)
        invstd = _empty((c,), F32, x)
        ws_bytes = _ws_bytes("ftx_bn_workspace_bytes", n, c)
        ws = _scratch(ws_bytes, x)
        _log_launch("bn_fwd", dict(n=n, c=c, reads=2 + (residual is not None), writes=1), lambda: check(L.ftx_bn_train_fwd(
            ptr(x), ptr(residual), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(momentum), float(eps),
            n, c, int(relu), ptr(y), ptr(mean), ptr(invstd), ptr(ws), ws_bytes, _stream_scratch()), "ftx_bn_train_fwd"))
        ctx.save_for_backward(x, y, gamma, beta, mean, invstd)
        ctx.relu = int(relu)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, beta, mean, invstd = ctx.saved_tensors
        gy = req(gy.contiguous(), F32, "bn grad", 2)
        gx, gres, ggamma, gbeta = _bn_backward_launch(gy, x, y, gamma, beta, mean, invstd, ctx.relu, ctx.has_res)
        return gx, gres, ggamma, gbeta, None, None, None, None, None


class _BatchNormEval(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, eps, relu):
        L = _lib.load()
        x = req(x.contiguous(), F32, "bn x", 2)
        n, c = x.shape
        if residual is not None:
            residual = req(residual.contiguous(), F32, "bn residual", 2)
        y = torch.empty_like(x)
        check(L.ftx_bn_eval_fwd(ptr(x), ptr(residual), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), n, c, int(relu),
                                ptr(y), stream()), "ftx_bn_eval_fwd")
        ctx.save_for_backward(y, gamma, running_var)
        ctx.eps, ctx.relu, ctx.has_res = float(eps), int(relu), residual is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        # Eval-mode gradients are elementwise; torch device ops (no library call needed).
        y, gamma, running_var = ctx.saved_tensors
        dy = gy * (y > 0) if ctx.relu else gy
        scale = gamma * torch.rsqrt(running_var + ctx.eps)
        gx = dy * scale
        return gx, (dy if ctx.has_res else None), None, None, None, None, None, None


def _remask_beta(beta):
    """beta for ftx_bn_train_bwd: with it the ReLU mask of a residual-free BatchNorm is recomputed from x instead of read from y
    (FTX_BN_REMASK=0 withholds it: A/B aid, read at call time)."""
    return 0 if os.environ.get("FTX_BN_REMASK") == "0" else ptr(beta)


def _bn_backward_launch(gy, x, y, gamma, beta, mean, invstd, relu, has_res):
    L = _lib.load()
    n, c = x.shape
    gx = torch.empty_like(x)
    gres = torch.empty_like(x) if has_res else None
    ggamma = _empty((c,), F32, x)
    gbeta = _empty((c,), F32, x)
    ws_bytes = _ws_bytes("ftx_bn_workspace_bytes", n, c)
    ws = _scratch(ws_bytes, x)
    # two passes (statistics, apply), each reading gy and x (and y for the ReLU mask when a residual went into it: otherwise the mask is
    # recomputed from x); one or two row matrices written
    _log_launch("bn_bwd", dict(n=n, c=c, reads=2 * (2 + (1 if (relu and has_res) else 0)), writes=1 + (1 if has_res else 0)), lambda: check(L.ftx_bn_train_bwd(
        ptr(gy), ptr(x), ptr(y), ptr(gamma), _remask_beta(beta), ptr(mean), ptr(invstd), n, c, int(relu), ptr(gx), ptr(gres), ptr(ggamma),
        ptr(gbeta), ptr(ws), ws_bytes, _stream_scratch()), "ftx_bn_train_bwd"))
    return gx, gres, ggamma, gbeta


class _ConvBNTrain(torch.autograd.Function):
    """spnn.Conv3d -> spnn.BatchNorm (training statistics) (-> + residual) (-> ReLU) as ONE autograd node
    (models/spvcnn.py:22-35,38-50,53-79).  The reduce pass of the convolution produces the BatchNorm's batch statistics while
    it writes the convolution output (ftx_spconv_reduce_stats: the last block to finish leaves the column totals), so that output is read
    once, by the one apply launch (ftx_bn_train_fwd_totals), instead of twice;
    one node instead of two also halves the host work per layer, and everything that does not outlive the call -- the pair rows `tmp`,
    the partial statistics, the BatchNorm input gradient between the two halves of the backward, the kernel workspaces -- lives in the
    stream's scratch buffer instead of six allocator round trips per layer and direction."""

    @staticmethod
    def forward(ctx, feats, kernel, km, transposed, residual, gamma, beta, running_mean, running_var, momentum, eps, relu):
        L = _lib.load()
        feats = req(feats.contiguous(), F32, "conv3d feats", 2)
        kernel = req(kernel.contiguous(), F32, "conv3d kernel", 3)
        kvol, ca, co, n_in, n_out = _conv_shapes(feats, kernel, km, transposed)
        if residual is not None:
            residual = req(residual.contiguous(), F32, "bn residual", 2)
            if residual.shape != (n_out, co):
                raise ValueError("conv_bn: residual shape mismatch")
        for t, nm in ((gamma, "gamma"), (beta, "beta")):
            req(t, F32, "bn " + nm, 1)
            if t.shape[0] != co:
                raise ValueError("conv_bn: BatchNorm parameter length != output channels")
        stats = _empty((2, co), F32, feats)              # row 0: batch mean, row 1: 1 / sqrt(var + eps)
        p_mean, p_invstd = stats.data_ptr(), stats.data_ptr() + 4 * co
        st = _stream_scratch()
        direct = (transposed and km.fine_bijective) or n_out == 0 or km.n_pairs == 0
        if direct:
            x = _conv_forward(feats, kernel, km, transposed)
            y = torch.empty_like(x)
            ws_bytes = _ws_bytes("ftx_bn_workspace_bytes", n_out, co)
            ws, = _carve(x, ws_bytes)
            _log_launch("bn_fwd", dict(n=n_out, c=co, reads=2 + (residual is not None), writes=1), lambda: check(L.ftx_bn_train_fwd(
                ptr(x), ptr(residual), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(momentum), float(eps),
                n_out, co, int(relu), ptr(y), p_mean, p_invstd, ws, ws_bytes, st), "ftx_bn_train_fwd"))
        elif not transposed and ostat_preferred(ca, co, kvol, rows=n_out):
            # thin layer: convolution + statistics in one launch, then the apply pass
            nb = _ws_bytes("ftx_spconv_ostat_blocks", n_out)
            part, = _carve(feats, 16 * (nb + 1) * co)
            x = _spconv_ostat(feats, kernel, km.nbr, n_out, co, 0, 0, part=part, nb=nb, pairs=km.n_pairs)
            y = torch.empty_like(x)
            _log_launch("bn_fwd", dict(n=n_out, c=co, reads=1 + (residual is not None), writes=1), lambda: check(L.ftx_bn_train_fwd_totals(
                ptr(x), ptr(residual), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(momentum),
                float(eps), n_out, co, int(relu), ptr(y), p_mean, p_invstd, part + 16 * nb * co, st), "ftx_bn_train_fwd_totals"))
        else:
            gather, pos = (km.pair_out, km.pos_t) if transposed else (km.pair_in, km.pos)
            x = _empty((n_out, co), F32, feats)
            y = torch.empty_like(x)
            nb = _ws_bytes("ftx_spconv_reduce_stats_blocks", n_out, co)
            tmp, part = _carve(feats, 4 * km.n_pairs * co, 16 * (nb + 1) * co)    # nb partial rows + the totals row, float64
            meta = dict(pairs=km.n_pairs, n_out=n_out, ca=ca, co=co, kvol=kvol)
            _log_launch("spconv_pairs_gemm", meta, lambda: check(L.ftx_spconv_pairs_gemm(
                ptr(feats), n_in, ptr(gather), ptr(kernel), 0, ptr(km.koff), km.n_pairs, ca, co, kvol, tmp, st), "ftx_spconv_pairs_gemm"))
            _log_launch("spconv_reduce", meta, lambda: check(L.ftx_spconv_reduce_stats(
                tmp, ptr(pos), n_out, co, kvol, ptr(x), part, nb, st), "ftx_spconv_reduce_stats"))
            _log_launch("bn_fwd", dict(n=n_out, c=co, reads=1 + (residual is not None), writes=1), lambda: check(L.ftx_bn_train_fwd_totals(
                ptr(x), ptr(residual), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(momentum),
                float(eps), n_out, co, int(relu), ptr(y), p_mean, p_invstd, part + 16 * nb * co, st), "ftx_bn_train_fwd_totals"))
        ctx.save_for_backward(feats, kernel, x, y, gamma, beta, stats)
        ctx.km, ctx.transposed, ctx.relu, ctx.has_res = km, transposed, int(relu), residual is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.load()
        feats, kernel, x, y, gamma, beta, stats = ctx.saved_tensors
        km, transposed = ctx.km, ctx.transposed
        gy = req(gy.contiguous(), F32, "conv_bn grad", 2)
        n, co = x.shape
        kvol, ca, _ = kernel.shape
        need_feats, need_kernel = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        st = _stream_scratch()
        p_mean, p_invstd = stats.data_ptr(), stats.data_ptr() + 4 * co
        gparams = _empty((2, co), F32, x)               # row 0: d gamma, row 1: d beta
        gres = torch.empty_like(x) if ctx.has_res else None
        bn_ws_bytes = _ws_bytes("ftx_bn_workspace_bytes", n, co)
        in_side, out_side = (km.pair_out, km.pair_in) if transposed else (km.pair_in, km.pair_out)
        direct = need_feats and (not transposed) and km.fine_bijective
        ostat = need_feats and (not direct) and (not transposed) and km.submanifold and km.n_pairs > 0 and ostat_preferred(co, ca, kvol, True, rows=feats.shape[0])
        n_feats = feats.shape[0]
        wg_bytes = _ws_bytes("ftx_spconv_pairs_wgrad_workspace_bytes", km.n_pairs, ca, co, kvol) if (need_kernel and km.n_pairs > 0) else 0
        tmp_bytes = 4 * km.n_pairs * ca if (need_feats and not direct and not ostat) else 0
        bn_ws, gx, tmp, wg_ws = _carve(x, bn_ws_bytes, 4 * n * co, tmp_bytes, wg_bytes)
        # BatchNorm half: gx = d loss / d (convolution output) stays in the scratch buffer, it is consumed by the two calls below
        _log_launch("bn_bwd", dict(n=n, c=co, reads=2 * (2 + (1 if (ctx.relu and ctx.has_res) else 0)), writes=1 + (1 if ctx.has_res else 0)), lambda: check(L.ftx_bn_train_bwd(
            ptr(gy), ptr(x), ptr(y), ptr(gamma), _remask_beta(beta), p_mean, p_invstd, n, co, ctx.relu, gx, ptr(gres), gparams.data_ptr(), gparams.data_ptr() + 4 * co,
            bn_ws, bn_ws_bytes, st), "ftx_bn_train_bwd"))
        g_feats = g_kernel = None
        meta = dict(pairs=km.n_pairs, n_out=n_feats, ca=co, co=ca, kvol=kvol)
        if need_feats:
            g_feats = _empty((n_feats, ca), F32, x)
            if km.n_pairs == 0 or n_feats == 0:
                g_feats.zero_()
            elif direct:
                _log_launch("spconv_pairs_gemm", dict(meta, direct=True), lambda: check(L.ftx_spconv_pairs_gemm_scatter(
                    gx, n, ptr(km.pair_out), ptr(km.pair_in), ptr(kernel), 1, ptr(km.koff), km.n_pairs, co, ca, kvol, ptr(g_feats), n_feats, st),
                    "ftx_spconv_pairs_gemm_scatter"))
            elif ostat:
                _log_launch("spconv_ostat", dict(meta, direct=True), lambda: check(L.ftx_spconv_ostat(
                    gx, n, ptr(km.nbr), n_feats, ptr(kernel), 1, 1, co, ca, kvol, ptr(g_feats), 0, 0, st), "ftx_spconv_ostat"))
            else:
                pos_in = km.pos if transposed else km.pos_t
                _log_launch("spconv_pairs_gemm", meta, lambda: check(L.ftx_spconv_pairs_gemm(
                    gx, n, ptr(out_side), ptr(kernel), 1, ptr(km.koff), km.n_pairs, co, ca, kvol, tmp, st), "ftx_spconv_pairs_gemm"))
                _log_launch("spconv_reduce", meta, lambda: check(L.ftx_spconv_reduce(tmp, ptr(pos_in), n_feats, ca, kvol, ptr(g_feats), st), "ftx_spconv_reduce"))
        if need_kernel:
            g_kernel = _empty((kvol, ca, co), F32, x)
            _log_launch("spconv_pairs_wgrad", dict(pairs=km.n_pairs, n_out=n, ca=ca, co=co, kvol=kvol), lambda: check(L.ftx_spconv_pairs_wgrad(
                ptr(feats), n_feats, ptr(in_side), gx, n, ptr(out_side), ptr(km.koff), km.n_pairs, ca, co, kvol, ptr(g_kernel), wg_ws, wg_bytes, st),
                "ftx_spconv_pairs_wgrad"))
        return g_feats, g_kernel, None, None, gres, gparams[0], gparams[1], None, None, None, None, None


def conv_bn_train(feats, kernel, km, transposed, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, residual=None, relu=False):
    """Conv3d -> BatchNorm(training) (+ residual) (+ ReLU) in one autograd node; see _ConvBNTrain."""
    return _ConvBNTrain.apply(feats, kernel, km, transposed, residual, gamma, beta, running_mean, running_var, momentum, eps, relu)


def batch_norm(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, residual=None, relu=False):
    """y = relu?(BN(x) (+ residual)) over the rows of x (N,C)."""
    if training:
        return _BatchNormTrain.apply(x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu)
    return _BatchNormEval.apply(x, residual, gamma, beta, running_mean, running_var, eps, relu)


# ---------------------------------------------------------------- 2D -> 3D lift, nearest resample
def lift_segments(img_idx, point_batch, b, gh, gw, H, W) -> Segments:
    """Points sorted by the grid cell they read: turns the lift's backward into a gather-reduce."""
    L = _lib.load()
    req(img_idx, I64, "lift img_idx", 2)
    req(point_batch, I32, "lift point_batch", 1)
    n = img_idx.shape[0]
    cells = _empty((n,), I32, point_batch)
    check(L.ftx_lift_cells(ptr(img_idx), ptr(point_batch), n, int(b), int(gh), int(gw), int(H), int(W), ptr(cells), stream()), "ftx_lift_cells")
    return Segments(cells, int(b) * int(gh) * int(gw))


class _LiftGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, img_idx, point_batch, H, W, seg):
        L = _lib.load()
        grid = req(grid.contiguous(), F32, "lift grid", 4)
        req(img_idx, I64, "lift img_idx", 2)
        req(point_batch, I32, "lift point_batch", 1)
        b, gh, gw, c = grid.shape
        n = img_idx.shape[0]
        if img_idx.shape != (n, 2) or point_batch.shape[0] != n:
            raise ValueError("lift_gather: img_idx must be (N,2), point_batch (N,)")
        out = _empty((n, c), F32, grid)
        check(L.ftx_lift_gather_fwd(ptr(grid), ptr(img_idx), ptr(point_batch), n, b, gh, gw, c, int(H), int(W), ptr(out), stream()), "ftx_lift_gather_fwd")
        ctx.save_for_backward(img_idx, point_batch)
        ctx.dims = (b, gh, gw, c, int(H), int(W))
        ctx.seg = seg
        return out

    @staticmethod
    def backward(ctx, go):
        L = _lib.load()
        img_idx, point_batch = ctx.saved_tensors
        b, gh, gw, c, H, W = ctx.dims
        go = req(go.contiguous(), F32, "lift grad", 2)
        gg = _empty((b, gh, gw, c), F32, go)
        seg = ctx.seg
        if seg is not None:
            if seg.m != b * gh * gw or seg.order.shape[0] != go.shape[0]:
                raise ValueError("lift_gather: segments do not match the grid / points")
            check(L.ftx_segment_sum(ptr(go), ptr(seg.order), ptr(seg.seg_off), go.shape[0], c, seg.m, ptr(gg), stream()), "ftx_segment_sum")
        else:
            check(L.ftx_lift_gather_bwd(ptr(go), ptr(img_idx), ptr(point_batch), go.shape[0], b, gh, gw, c, H, W, ptr(gg), stream()), "ftx_lift_gather_bwd")
        return gg, None, None, None, None, None


def lift_gather(grid, img_idx, point_batch, H, W, seg=None):
    """Per-point rows of nearest-upsample(grid -> (H,W)) without materialising the map.
    `seg` = lift_segments(...) makes the backward an atomic-free, bit-reproducible gather-reduce."""
    return _LiftGather.apply(grid, img_idx, point_batch, H, W, seg)


class _ResampleNearest(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        L = _lib.load()
        x = req(x.contiguous(), F32, "resample x", 4)
        b, c, ih, iw = x.shape
        out = _empty((b, c, int(oh), int(ow)), F32, x)
        check(L.ftx_resample_nearest_fwd(ptr(x), b, c, ih, iw, int(oh), int(ow), ptr(out), stream()), "ftx_resample_nearest_fwd")
        ctx.dims = (b, c, ih, iw, int(oh), int(ow))
        return out

    @staticmethod
    def backward(ctx, go):
        L = _lib.load()
        b, c, ih, iw, oh, ow = ctx.dims
        go = req(go.contiguous(), F32, "resample grad", 4)
        gi = _empty((b, c, ih, iw), F32, go)
        check(L.ftx_resample_nearest_bwd(ptr(go), b, c, ih, iw, oh, ow, ptr(gi), stream()), "ftx_resample_nearest_bwd")
        return gi, None, None


def resample_nearest(x, size):
    """nn.Upsample(size) (nearest) on NCHW."""
    return _ResampleNearest.apply(x, size[0], size[1])


# ---------------------------------------------------------------- LayerNorm (+ the residual add in front of it)
_FUSED_LN = os.environ.get("FTX_FUSED_LN", "1") != "0"      # A/B aid: 0 = torch's add + LayerNorm kernels in the ViT blocks


def layer_norm_supported(x: torch.Tensor) -> bool:
    return _FUSED_LN and x.is_cuda and x.dtype == F32 and x.shape[-1] % 256 == 0 and 256 <= x.shape[-1] <= 1024


def _ln_forward(x, y, weight, bias, eps, y_bias=None):
    L = _lib.load()
    shape = x.shape
    c = shape[-1]
    x2 = req(x.contiguous().view(-1, c), F32, "layer_norm x", 2)
    y2 = req(y.contiguous().view(-1, c), F32, "layer_norm y", 2) if y is not None else None
    if y2 is not None and y2.shape != x2.shape:
        raise ValueError("add_layer_norm: x and y differ in shape")
    for t, nm in ((weight, "weight"), (bias, "bias"), (y_bias, "y_bias")):
        if t is None and nm == "y_bias":
            continue
        req(t, F32, "layer_norm " + nm, 1)
        if t.shape[0] != c:
            raise ValueError("layer_norm: parameter length != row length")
    if y_bias is not None and y2 is None:
        raise ValueError("add_layer_norm: y_bias without y")
    rows = x2.shape[0]
    h = torch.empty_like(x2)
    s = torch.empty_like(x2) if y2 is not None else x2
    stats = _empty((2, rows), F32, x2)               # row 0: mean, row 1: 1 / sqrt(var + eps)
    check(L.ftx_add_layernorm_fwd(ptr(x2), ptr(y2), ptr(y_bias), ptr(weight), ptr(bias), float(eps), rows, c, ptr(s) if y2 is not None else 0, ptr(h),
                                  stats.data_ptr(), stats.data_ptr() + 4 * rows, stream()), "ftx_add_layernorm_fwd")
    return s, h, stats, shape


def _ln_backward(gh, gs, s, weight, stats, with_y_bias=False):
    L = _lib.load()
    rows, c = s.shape
    gh = req(gh.contiguous().view(rows, c), F32, "layer_norm grad", 2)
    gs = req(gs.contiguous().view(rows, c), F32, "layer_norm residual grad", 2) if gs is not None else None
    gx = torch.empty_like(s)
    gparams = _empty((3 if with_y_bias else 2, c), F32, s)      # d gamma, d beta (, d y_bias)
    ws_bytes = _ws_bytes("ftx_layernorm_bwd_workspace_bytes", rows, c)
    ws = _scratch(ws_bytes, s)
    check(L.ftx_add_layernorm_bwd(ptr(gh), ptr(gs), ptr(s), ptr(weight), stats.data_ptr(), stats.data_ptr() + 4 * rows, rows, c, int(with_y_bias),
                                  ptr(gx), ptr(gparams), ptr(ws), ws_bytes, stream()), "ftx_add_layernorm_bwd")
    return gx, gparams


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        s, h, stats, shape = _ln_forward(x, None, weight, bias, eps)
        ctx.save_for_backward(s, weight, stats)
        ctx.shape = shape
        return h.view(shape)

    @staticmethod
    def backward(ctx, gh):
        s, weight, stats = ctx.saved_tensors
        gx, gparams = _ln_backward(gh, None, s, weight, stats)
        return gx.view(ctx.shape), gparams[0], gparams[1], None


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, y_bias, weight, bias, eps):
        s, h, stats, shape = _ln_forward(x, y, weight, bias, eps, y_bias)
        ctx.save_for_backward(s, weight, stats)
        ctx.shape, ctx.with_y_bias = shape, y_bias is not None
        return s.view(shape), h.view(shape)

    @staticmethod
    def backward(ctx, gs, gh):
        s, weight, stats = ctx.saved_tensors
        if gh is None:                                   # the normalised output was not used: only the sum's gradient passes
            g = gs if gs is not None else torch.zeros(ctx.shape, dtype=F32, device=s.device)
            gyb = colsum(g.reshape(-1, g.shape[-1])) if ctx.with_y_bias else None
            return g, g, gyb, None, None, None
        gx, gparams = _ln_backward(gh, gs, s, weight, stats, ctx.with_y_bias)
        gx = gx.view(ctx.shape)
        return gx, gx, (gparams[2] if ctx.with_y_bias else None), gparams[0], gparams[1], None


def layer_norm(x, weight, bias, eps=1e-5):
    """nn.LayerNorm over the last dimension (256 / 512 / 768 / 1024 floats per row)."""
    return _LayerNorm.apply(x, weight, bias, eps)


def add_layer_norm(x, y, weight, bias, eps=1e-5, y_bias=None):
    """(s, LayerNorm(s)) with s = x + y in one pass; the backward returns one gradient for both addends: the residual gradient plus the
    LayerNorm's input gradient, written once.  `y_bias`: y is a Linear's output computed WITHOUT its bias, s = x + (y + y_bias); the
    bias gradient then comes out of the same backward pass (no column-sum launches for that Linear)."""
    return _AddLayerNorm.apply(x, y, y_bias, weight, bias, eps)


# ---------------------------------------------------------------- column sums (bias gradients)
def colsum(x: torch.Tensor) -> torch.Tensor:
    """x (rows, cols) float32 -> (cols,) column sums (float64 accumulation, fixed order): the bias gradient of a Linear."""
    L = _lib.load()
    x = req(x.contiguous(), F32, "colsum x", 2)
    rows, cols = x.shape
    out = _empty((cols,), F32, x)
    ws_bytes = _ws_bytes("ftx_colsum_workspace_bytes", rows, cols)
    ws = _scratch(ws_bytes, x)
    check(L.ftx_colsum(ptr(x), rows, cols, ptr(out), ptr(ws), ws_bytes, stream()), "ftx_colsum")
    return out


# ---------------------------------------------------------------- ViT self-attention
class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, scale, tiling=(0, 0)):
        L = _lib.load()
        qkv = req(qkv.contiguous(), F32, "attention qkv", 5)
        b, t, three, h, d = qkv.shape
        if three != 3 or d != 64:
            raise ValueError(f"attention: qkv must be (B, T, 3, heads, 64), got {tuple(qkv.shape)}")
        out = _empty((b, t, h * d), F32, qkv)
        lse = _empty((b, h, t), F32, qkv)
        qw, split = int(tiling[0]), int(tiling[1])
        _log_launch("attn_fwd", dict(b=b, t=t, h=h, d=d, products=2), lambda: check(L.ftx_attn_fwd_tiled(
            ptr(qkv), b, t, h, d, float(scale), ptr(out), ptr(lse), qw, split, stream()), "ftx_attn_fwd"))
        ctx.save_for_backward(qkv, out, lse)
        ctx.scale, ctx.tiling = float(scale), (qw, split)
        return out

    @staticmethod
    def backward(ctx, go):
        L = _lib.load()
        qkv, out, lse = ctx.saved_tensors
        b, t, _, h, d = qkv.shape
        go = req(go.contiguous(), F32, "attention grad", 3)
        gqkv = torch.empty_like(qkv)
        ws_bytes = int(L.ftx_attn_bwd_workspace_bytes(b, t, h))
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=qkv.device)
        _log_launch("attn_bwd", dict(b=b, t=t, h=h, d=d, products=7), lambda: check(L.ftx_attn_bwd_tiled(
            ptr(qkv), ptr(out), ptr(go), ptr(lse), b, t, h, d, ctx.scale, ptr(gqkv), ptr(ws), ws_bytes, ctx.tiling[0], ctx.tiling[1], stream()), "ftx_attn_bwd"))
        return gqkv, None, None


def attention(qkv, scale, tiling=(0, 0)):
    """softmax(Q K^T * scale) V for qkv (B, T, 3, heads, 64) -> (B, T, heads*64).  `tiling` = (waves per block, key groups) of the
    kernels, (0, 0) = chosen per launch (ftx_attn_fwd_tiled in include/ftx.h): a per-call argument for tests and tools."""
    return _Attention.apply(qkv, scale, tiling)


# ---------------------------------------------------------------- fused sample_down
class _SampleDown(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, conv_w, conv_b, gamma, beta, running_mean, running_var, momentum, eps, training, oh, ow):
        L = _lib.load()
        img = req(img.contiguous(), F32, "sample_down img", 4)
        b, c, h, w = img.shape
        if c != 3 or conv_w.numel() != 9:
            raise ValueError("sample_down: the fused kernel is the 3 -> 3 channel BilinearModule of the reference")
        conv_w = req(conv_w.contiguous().view(3, 3), F32, "sample_down conv weight", 2)
        out = _empty((b, 3, int(oh), int(ow)), F32, img)
        saved = torch.empty((33,), dtype=torch.float64, device=img.device)
        ws_bytes = int(L.ftx_sample_down_workspace_bytes())
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=img.device)
        check(L.ftx_sample_down_fwd(ptr(img), b, h, w, int(oh), int(ow), ptr(conv_w), ptr(conv_b), ptr(gamma), ptr(beta), ptr(running_mean),
                                    ptr(running_var), float(momentum), float(eps), int(bool(training)), ptr(out), ptr(saved), ptr(ws), ws_bytes, stream()),
              "ftx_sample_down_fwd")
        ctx.save_for_backward(img, conv_w, conv_b, gamma, saved)
        ctx.training = bool(training)
        ctx.dims = (b, h, w, int(oh), int(ow))
        return out

    @staticmethod
    def backward(ctx, go):
        L = _lib.load()
        if not ctx.training:
            raise RuntimeError("sample_down: backward is implemented for training-mode statistics only")
        img, conv_w, conv_b, gamma, saved = ctx.saved_tensors
        b, h, w, oh, ow = ctx.dims
        go = req(go.contiguous(), F32, "sample_down grad", 4)
        gw = _empty((3, 3), F32, go)
        gb, gg, gbeta = _empty((3,), F32, go), _empty((3,), F32, go), _empty((3,), F32, go)
        ws_bytes = int(L.ftx_sample_down_workspace_bytes())
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=go.device)
        check(L.ftx_sample_down_bwd(ptr(img), ptr(go), b, h, w, oh, ow, ptr(conv_w), ptr(conv_b), ptr(gamma), ptr(saved), ptr(gw), ptr(gb), ptr(gg),
                                    ptr(gbeta), ptr(ws), ws_bytes, stream()), "ftx_sample_down_bwd")
        return None, gw, gb, gg, gbeta, None, None, None, None, None, None, None


def sample_down(img, conv_w, conv_b, gamma, beta, running_mean, running_var, momentum, eps, training, size):
    """Conv1x1(3->3) + ReLU + BatchNorm2d + nearest pick, fused (image_models_billinear.py:8-24)."""
    return _SampleDown.apply(img, conv_w, conv_b, gamma, beta, running_mean, running_var, momentum, eps, training, size[0], size[1])


# ---------------------------------------------------------------- fused losses + metric
class _FusionLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, l3, l2, l3b, l2b, label, class_weights, lambda_xm, conf3d, conf2d, ignore_index, ce_scale=1.0):
        L = _lib.load()
        l3 = req(l3.contiguous(), F32, "loss lidar_seg_logit", 2)
        l2 = req(l2.contiguous(), F32, "loss img_seg_logit", 2)
        n, c = l3.shape
        if l2.shape != (n, c) or label.shape[0] != n:
            raise ValueError("fusion_loss: shape mismatch")
        dual = l3b is not None
        if dual:
            l3b = req(l3b.contiguous(), F32, "loss lidar_seg_logit2", 2)
            l2b = req(l2b.contiguous(), F32, "loss img_seg_logit2", 2)
        label = req(label.contiguous(), I64, "loss label", 1)
        losses = _empty((2,), F32, l3)
        g3, g2 = torch.empty_like(l3), torch.empty_like(l2)
        g3b = torch.empty_like(l3) if dual else None
        g2b = torch.empty_like(l2) if dual else None
        ws_bytes = int(L.ftx_fusion_loss_workspace_bytes())
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=l3.device)
        check(L.ftx_fusion_loss_mix(ptr(l3), ptr(l2), ptr(l3b), ptr(l2b), ptr(label), ptr(class_weights), float(ce_scale), float(lambda_xm), n, c,
                                    int(ignore_index), ptr(losses), ptr(g3), ptr(g2), ptr(g3b), ptr(g2b), ptr(conf3d), ptr(conf2d), ptr(ws), ws_bytes,
                                    stream()), "ftx_fusion_loss_mix")
        ctx.save_for_backward(g3, g2, g3b, g2b) if dual else ctx.save_for_backward(g3, g2)
        ctx.dual = dual
        return losses

    @staticmethod
    def backward(ctx, g):
        # the kernel produced d(loss_2d + loss_3d); loss_2d depends only on (l2, l2b), loss_3d only on (l3, l3b)
        if ctx.dual:
            g3, g2, g3b, g2b = ctx.saved_tensors
            return g3 * g[1], g2 * g[0], g3b * g[1], g2b * g[0], None, None, None, None, None, None, None
        g3, g2 = ctx.saved_tensors
        if not torch.equal(g[0], g[1]):
            raise RuntimeError("fusion_loss (single head): loss_2d and loss_3d share logits; call backward on their sum")
        return g3 * g[0], g2 * g[0], None, None, None, None, None, None, None, None, None


def fusion_loss(preds, seg_label, class_weights, lambda_xm, dual_head, conf3d=None, conf2d=None, ignore_index=0, mix="additive"):
    """(loss_2d, loss_3d) in one fused pass; conf3d / conf2d (C,C) int64 tensors, when given, accumulate the SegIoU confusion
    matrices of models/metric.py:37-58.  mix="additive": CE + lambda*KL (SemanticTrainer.py:158-178); mix="torchpack":
    (1-lambda)*CE + lambda*KL when lambda > 0 (modules/SemanticTorchpackTrainer.py:70-106)."""
    if mix not in ("additive", "torchpack"):
        raise ValueError("fusion_loss: mix must be 'additive' or 'torchpack'")
    ce_scale = (1.0 - float(lambda_xm)) if (mix == "torchpack" and lambda_xm > 0) else 1.0
    out = _FusionLoss.apply(preds["lidar_seg_logit"], preds["img_seg_logit"], preds["lidar_seg_logit2"] if dual_head else None,
                            preds["img_seg_logit2"] if dual_head else None, seg_label.long(), class_weights, lambda_xm, conf3d, conf2d, ignore_index,
                            ce_scale)
    return out[0], out[1]


def eval_scatter_back(logits3d, logits2d, inverse, gt, class_labels, conf3d=None, conf2d=None, conf_ens=None, want_preds=True):
    """Predictions of the model points mapped to the original points + confusion-matrix update in one kernel
    (reference data/utils/validate.py:62-120, data/utils/evaluate.py:12-26; see include/ftx.h).

    inverse (M,) int64: row of the model point of every original point (frame offset already added); gt (M,)
    learning ids; class_labels (C,) original id of every learning id.  conf_* are (C,C) int64 accumulators.
    Returns (pred_3d, pred_2d, pred_ens) in original label ids (None where not computed)."""
    L = _lib.load()
    ref = logits3d if logits3d is not None else logits2d
    if ref is None:
        raise ValueError("eval_scatter_back needs at least one logits tensor")
    for t, name in ((logits3d, "logits3d"), (logits2d, "logits2d")):
        if t is not None:
            req(t, F32, "eval_scatter_back " + name, 2)
    n, c = ref.shape
    if logits3d is not None and logits2d is not None and logits3d.shape != logits2d.shape:
        raise ValueError("eval_scatter_back: logits shapes differ")
    req(inverse, I64, "eval_scatter_back inverse", 1)
    gt = gt.to(I32).contiguous()
    req(gt, I32, "eval_scatter_back gt", 1)
    class_labels = class_labels.to(device=ref.device, dtype=I32).contiguous()
    if class_labels.numel() != c or gt.shape[0] != inverse.shape[0]:
        raise ValueError("eval_scatter_back: class_labels must have one id per class and gt one label per original point")
    m = inverse.shape[0]
    for t, name in ((conf3d, "conf3d"), (conf2d, "conf2d"), (conf_ens, "conf_ens")):
        if t is not None:
            req(t, I64, "eval_scatter_back " + name, 2)
            if tuple(t.shape) != (c, c):
                raise ValueError("eval_scatter_back: %s must be (%d, %d)" % (name, c, c))
    mk = (lambda cond: _empty((m,), I32, ref) if (want_preds and cond) else None)
    p3, p2, pe = mk(logits3d is not None), mk(logits2d is not None), mk(logits3d is not None and logits2d is not None)
    bad = torch.zeros((1,), dtype=I32, device=ref.device)
    check(L.ftx_eval_scatter_back(ptr(logits3d), ptr(logits2d), n, c, ptr(inverse), ptr(gt), m, ptr(class_labels), ptr(p3), ptr(p2), ptr(pe),
                                  ptr(conf3d), ptr(conf2d), ptr(conf_ens), ptr(bad), stream()), "ftx_eval_scatter_back")
    return p3, p2, pe, bad
