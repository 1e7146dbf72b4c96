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
    __slots__ = ("nbr", "pos", "pos_t", "pair_in", "pair_out", "koff", "n_pairs", "n_in", "n_out", "out_coords", "kvol", "fine_bijective", "submanifold")

    def __init__(self, nbr, pos, pos_t, pair_in, pair_out, koff, n_pairs, n_in, n_out, out_coords, fine_bijective=False, submanifold=False):
        self.nbr, self.pos, self.pos_t, self.pair_in, self.pair_out, self.koff = nbr, pos, pos_t, pair_in, pair_out, koff
        self.n_pairs, self.n_in, self.n_out, self.out_coords = n_pairs, n_in, n_out, out_coords
        self.kvol = nbr.shape[0]
        # strided 2^3 map: every input (fine) voxel has exactly one (parent, offset), so pair_in is a permutation of the fine rows
        self.fine_bijective = bool(fine_bijective) and n_pairs == n_in
        # stride-1 odd kernel on one coordinate set: nbr[k, o] = i <=> nbr[K-1-k, i] = o, so the data gradient can read the same table mirrored
        self.submanifold = bool(submanifold) and n_in == n_out and self.kvol % 2 == 1


class CoordinateManager:
    """Hash tables and kernel maps of one batch, built once and shared by every
    layer of the level (the reference caches them on the tensors:
    models/utils.py:60-61 copies `kernel_maps`, torchsparse conv3d fills them)."""

    def __init__(self):
        self.tables = {}       # stride -> HashTable over sphash(coords at that stride)
        self.coords = {}       # stride -> (N,4) int32
        self.kernel_maps = {}  # (ks, cur_stride, stride) -> KernelMap
        self._offsets = {}
        # set by initial_voxelize(levels=...): the floored point coordinates and, per stride, (sorted unique hashes, first-occurrence
        # point rows) of that level -- every level's coordinates and hash table then come without a host read (_materialize_level)
        self.points = None
        self.level_data = None

    def offsets(self, ks, stride, device):
        key = (ks, stride)
        if key not in self._offsets:
            self._offsets[key] = torch.from_numpy(Fn.kernel_offsets(ks, stride)).to(device)
        return self._offsets[key]

    def table(self, stride):
        if stride not in self.tables:
            self.tables[stride] = Fn.HashTable(Fn.sphash(self.coords[stride]))
        return self.tables[stride]

    def _materialize_level(self, stride):
        """Coordinates and hash table of a level that initial_voxelize(levels=...) already found: no sort, no host read."""
        if stride in self.coords or not self.level_data or stride not in self.level_data:
            return stride in self.coords
        hashes, first = self.level_data[stride][:2]
        self.coords[stride] = Fn.level_coords(self.points, first, stride)
        if stride not in self.tables:
            self.tables[stride] = Fn.HashTable(hashes)      # == sphash(coords[stride]): the level's rows are in hash order
        return True

    def kernel_map(self, ks, cur_stride, stride) -> KernelMap:
        key = (ks, cur_stride, stride)
        km = self.kernel_maps.get(key)
        if km is not None:
            return km
        return drain(self.kernel_map_steps(ks, cur_stride, stride))

    def kernel_map_steps(self, ks, cur_stride, stride):
        """kernel_map() as a generator: it yields "sync" after the kernels whose result sizes the next
        arrays have been launched and before that size is read back, so a scheduler can issue other
        work (the image branch) instead of blocking on the read (models/_fusion_common.run_fusion)."""
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
            if new_stride not in self.coords and not self._materialize_level(new_stride):
                yield from self.downsample_steps(cur_stride, stride)
            out_coords = self.coords[new_stride]
        nbr = Fn.kernel_map_build(out_coords, off, table)
        pos, koff = Fn.kernel_map_count(nbr)
        # a strided kernel-2 map joins every input voxel to exactly one (parent, offset): P = N_in;
        # for the submanifold maps the pair count is data dependent (one host read)
        if ks == stride and ks == 2:
            n_pairs = n_in
        else:
            pending = HostRead(koff[-1:])
            yield "sync"
            n_pairs = pending.value()
        return self._finish_map(key, nbr, pos, koff, n_pairs, n_in, out_coords)

    def _finish_map(self, key, nbr, pos, koff, n_pairs, n_in, out_coords):
        pos, pos_t, pair_in, pair_out = Fn.kernel_map_pairs(nbr, pos, n_in, n_pairs)
        ks, cur_stride, stride = key
        km = KernelMap(nbr, pos, pos_t, pair_in, pair_out, koff, n_pairs, n_in, out_coords.shape[0], out_coords,
                       fine_bijective=(ks == stride == 2), submanifold=(stride == 1 and ks % 2 == 1))
        self.kernel_maps[key] = km
        return km

    def downsample_steps(self, cur_stride, stride):
        """Coordinates of level cur_stride * stride (spdownsample: floor, unique in hash order)."""
        launched = self._downsample_launch(cur_stride, stride)
        yield "sync"
        self._downsample_finish(cur_stride * stride, *launched)

    def _downsample_launch(self, cur_stride, stride):
        down = Fn.downsample_coords(self.coords[cur_stride], cur_stride * stride)
        uniq, first, cnt = Fn.unique_sorted(Fn.sphash(down))
        return down, first, HostRead(cnt)

    def _downsample_finish(self, new_stride, down, first, pending):
        n_out = pending.value()  # the one host read per level: sizes the next level's tensors
        self.coords[new_stride] = Fn.gather_coords(down, first[:n_out].contiguous())

    def unet_levels_steps(self, strides, ks=3, down_ks=2):
        """Everything a U-Net over `strides` will ask for -- the kernel-`ks` submanifold map of every level and
        the strided kernel-`down_ks` map between consecutive levels -- built up front.

        When the levels were found by initial_voxelize(levels=...) every level's coordinates are already known, so the neighbour
        tables of ALL levels are built first and their data-dependent pair counts come back in ONE host read (after a single "sync"
        yield).  Otherwise (lazy form) the two data-dependent sizes of a level -- its pair count and the voxel count of the next level
        -- come back in one read per level."""
        if self.level_data and all(s in self.level_data for s in strides):
            pend = []
            for s in strides:
                self._materialize_level(s)
                coords = self.coords[s]
                nbr = Fn.kernel_map_build(coords, self.offsets(ks, s, coords.device), self.table(s))
                pos, koff = Fn.kernel_map_count(nbr)
                pend.append((s, coords, nbr, pos, koff))
            pending = HostRead(torch.cat([koff[-1:] for _, _, _, _, koff in pend]))
            yield "sync"
            for (s, coords, nbr, pos, koff), n_pairs in zip(pend, pending.values()):
                self._finish_map((ks, s, 1), nbr, pos, koff, n_pairs, coords.shape[0], coords)
            for s, nxt in zip(strides[:-1], strides[1:]):
                drain(self.kernel_map_steps(down_ks, s, nxt // s))   # no host read: P = N_in, the next level exists
            return
        for i, s in enumerate(strides):
            coords = self.coords[s]
            n_in = coords.shape[0]
            key = (ks, s, 1)
            nbr = Fn.kernel_map_build(coords, self.offsets(ks, s, coords.device), self.table(s))
            pos, koff = Fn.kernel_map_count(nbr)
            last = i + 1 == len(strides)
            if last:
                pending = HostRead(koff[-1:])
                yield "sync"
                self._finish_map(key, nbr, pos, koff, pending.value(), n_in, coords)
                break
            ratio = strides[i + 1] // s
            down = Fn.downsample_coords(coords, s * ratio)
            uniq, first, cnt = Fn.unique_sorted(Fn.sphash(down))
            pending = HostRead(torch.cat([koff[-1:].to(cnt.dtype), cnt]))
            yield "sync"
            n_pairs, n_out = pending.values()
            self._finish_map(key, nbr, pos, koff, n_pairs, n_in, coords)
            self.coords[s * ratio] = Fn.gather_coords(down, first[:n_out].contiguous())
            drain(self.kernel_map_steps(down_ks, s, ratio))   # no host read: P = N_in


class HostRead:
    """A few device integers on their way to the host: the copy is queued behind the kernels that produce
    them when the object is made, `value()` waits for that copy only."""

    latest = None      # the read made last on this thread of issue (PendingIndex looks at it to see what a parked build waits for)

    def __init__(self, dev: torch.Tensor):
        self.host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
        self.host.copy_(dev, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        HostRead.latest = self

    def ready(self):
        return self.event.query()

    def values(self):
        self.event.synchronize()
        return [int(v) for v in self.host.tolist()]

    def value(self):
        return self.values()[0]


def drain(steps):
    """Run a `*_steps` generator to its end (blocking on every host read) and return its value."""
    while True:
        try:
            next(steps)
        except StopIteration as done:
            return done.value


_INDEX_STREAMS = {}


def index_stream(device):
    """The HIP stream coordinate structures are built on when they are built ahead of the forward (SPVCNN.prepare)."""
    key = (device.type, device.index)
    if key not in _INDEX_STREAMS:
        _INDEX_STREAMS[key] = torch.cuda.Stream(device=device)
    return _INDEX_STREAMS[key]


def _device_tensors(obj, seen):
    """Every CUDA tensor reachable from `obj` through dicts, sequences and plain objects (each once)."""
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        if obj.is_cuda:
            yield obj
        return
    if isinstance(obj, dict):
        for v in obj.values():
            yield from _device_tensors(v, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _device_tensors(v, seen)
    elif hasattr(obj, "__dict__") or hasattr(obj, "__slots__"):
        for name in list(getattr(obj, "__dict__", {})) + list(getattr(obj, "__slots__", ())):
            v = getattr(obj, name, None)
            if v is not None and not isinstance(v, (int, float, str, bool, torch.cuda.Event)):
                yield from _device_tensors(v, seen)


class PreparedIndex:
    """Coordinate structures of one batch built ahead of its forward, on the index stream: the PointTensor (with its point <-> voxel
    index caches) and the level-0 SparseTensor (with its CoordinateManager)."""

    def __init__(self, z, x0, event):
        self.z, self.x0, self.event = z, x0, event

    def take(self, stream):
        """Hand the structures to `stream`: it waits for the build, and every tensor is marked as used there, so the caching allocator
        (which owns them on the index stream) does not recycle one while a kernel of the consumer is still reading it."""
        if stream is not None:
            stream.wait_event(self.event)
            for t in _device_tensors((self.z, self.x0), set()):
                t.record_stream(stream)
        return self.z, self.x0


class PendingIndex:
    """An index build in flight on the index stream: a `*_steps` generator that has been issued up to one of its host reads.  step()
    resumes it -- blocking on that read, then issuing up to the next one -- with the index stream, the device and the grad mode of the
    build in force for exactly the duration of the call (a generator suspended inside a `with torch.cuda.stream(...)` block would leave
    that stream current for its caller).  `done` is the PreparedIndex once the build has been issued to its end."""

    def __init__(self, steps, stream, device, grad):
        self.steps, self.stream, self.device, self.grad = steps, stream, device, grad
        self.done = None
        self.read = None       # the HostRead the build is parked on

    def ready(self):
        """True when step() would not block: the read the build is parked on has arrived."""
        return self.done is None and self.read is not None and self.read.ready()

    def step(self):
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream), torch.set_grad_enabled(self.grad):
            try:
                HostRead.latest = None
                next(self.steps)
                self.read = HostRead.latest
                return False
            except StopIteration as fin:
                z, x0 = fin.value
                ev = torch.cuda.Event()
                ev.record()
                self.done = PreparedIndex(z, x0, ev)
                return True


class SparseTensor:
    def __init__(self, feats, coords, stride=1):
        self.F = feats
        self.C = coords
        self.s = stride
        self.coord_maps = {}
        self.kernel_maps = {}
        self.cm = None  # CoordinateManager, attached by initial_voxelize
        self.prepared = None  # PreparedIndex / PendingIndex, attached by SPVCNN.prepare (coordinate structures built ahead of the forward)

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
