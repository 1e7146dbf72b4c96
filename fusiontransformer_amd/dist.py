"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

Counterpart of the reference's DDP wiring (modules/TorchpackInterface.py:44-81:
`dist.init()` + `DistributedDataParallel(find_unused_parameters=True)`).  Frames
are independent units (the batch index is the 4th coordinate column,
data/collate.py:41-42), so ranks share nothing in the forward; the only exchange
is the gradient average, ~108 M fp32 = 432 MB per step.

Design for xGMI (point-to-point links, no switch): few large buckets (default
64 MB) so each all-reduce amortises its launch and RCCL can spread it over all 7
links; each bucket is one flat buffer: when its last gradient has arrived the
gradients are packed into it by ONE multi-tensor copy on the exchange stream,
reduced in place, and `p.grad` become views of it (no unpack); the all-reduce is
issued from the autograd hook of the bucket's last gradient, on a stream of its
own, and runs while the rest of the backward continues.  Buckets are launched strictly in
bucket order on every rank (collectives must match across ranks); the order is
rebuilt after the first step from the gradient-ready order rank 0 observed
(broadcast once, so every rank cuts identical buckets), so later steps overlap.  Parameters that never receive a gradient are frozen at model
construction instead of using `find_unused_parameters`.  BatchNorm statistics
stay per replica, as in the reference (no SyncBN anywhere)."""
from __future__ import annotations

import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL / cross-process device memory on this driver

import torch
import torch.distributed as dist


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def init_process_group(backend=None, force=False):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the env (torchrun contract).

    `force`: create the process group even at world size 1 (a one-rank RCCL communicator), so that the collective path of
    GradReducer(force_collectives=True) -- communicator, `async_op` all-reduces issued from the autograd hooks, `work.wait()` --
    runs on a single GPU exactly as it does on eight."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("FTX_FORCE_DEVICE", local_rank)))
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class _Bucket:
    __slots__ = ("params", "flat", "views", "pending", "work", "launched", "streams")

    def __init__(self, params, flat, views):
        self.params, self.flat, self.views = params, flat, views
        self.pending, self.work, self.launched, self.streams = len(params), None, False, {}


class GradReducer:
    """Bucketed, overlapped gradient all-reduce (average) for `model`'s trainable parameters."""

    def __init__(self, model, bucket_mb: float = 64.0, process_group=None, broadcast_params=True, force_collectives=False):
        """force_collectives: issue the broadcasts and all-reduces even at world size 1 (needs an initialised process group,
        init_process_group(force=True)); the result is the identity, the code path is the N > 1 one."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if force_collectives and not dist.is_initialized():
            raise RuntimeError("GradReducer(force_collectives=True) needs an initialised process group (init_process_group(force=True))")
        self.active = self.world > 1 or bool(force_collectives)     # collectives are issued
        self.bucket_bytes = int(bucket_mb * 1024 * 1024)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.step_idx = 0
        self._comm = {}
        self.ready_order = []
        self._rebuilt = False
        self._record = False
        if self.active and broadcast_params:
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0, group=self.pg)
        self._build(list(reversed(self.params)))
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad_ready)
        # The hooks keep the AccumulateGrad nodes (created on the default stream) alive while the model's
        # branches run on their own streams; autograd then orders each accumulation after its producer
        # stream, which is exactly what the bucket events rely on.  The advisory warning about it is noise.
        try:
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        except AttributeError:
            pass

    # -- bucket construction -------------------------------------------------
    def _build(self, ordered):
        self.buckets, self.bucket_of = [], {}
        cur, cur_bytes = [], 0
        groups = []
        for p in ordered:
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > self.bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for g in groups:
            pad4 = lambda n: (n + 3) & ~3     # every gradient view starts on a 16-byte boundary (vector loads in ftx_adam_step)
            flat = torch.zeros(sum(pad4(p.numel()) for p in g), dtype=g[0].dtype, device=g[0].device)
            off, views = 0, []
            for p in g:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += pad4(p.numel())
            b = _Bucket(g, flat, views)
            for p in g:
                self.bucket_of[p] = len(self.buckets)
            self.buckets.append(b)
        self.next_to_launch = 0

    def _agree_on_order(self, order):
        """Every rank must cut the SAME buckets (collectives are matched by issue order), but the gradient-ready order a rank
        observed in step 0 depends on how its two branch streams and its autograd thread interleaved: rank 0's order is
        broadcast and adopted by all (one small int64 broadcast, once per job)."""
        if not self.active:
            return order
        index = {p: i for i, p in enumerate(self.params)}
        ids = torch.tensor([index[p] for p in order], dtype=torch.int64)
        if dist.get_backend(self.pg) == "nccl":
            ids = ids.to(self.params[0].device)
        dist.broadcast(ids, src=0, group=self.pg)
        ids = ids.cpu().tolist()
        if sorted(ids) != list(range(len(self.params))):
            raise RuntimeError("GradReducer: the bucket order received from rank 0 is not a permutation of this rank's parameters")
        return [self.params[i] for i in ids]

    # -- per-step protocol ---------------------------------------------------
    def begin_step(self):
        if self.step_idx == 1 and not self._rebuilt and self.ready_order:
            seen = set(self.ready_order)
            order = self.ready_order + [p for p in reversed(self.params) if p not in seen]
            self._build(self._agree_on_order(order))
            self._rebuilt = True
        for b in self.buckets:
            b.pending, b.work, b.launched, b.streams = len(b.params), None, False, {}
        for p in self.params:
            p.grad = None    # autograd then MOVES each gradient into place (no add kernel per parameter); _launch packs the bucket
        self.next_to_launch = 0
        self._record = self.step_idx == 0
        if self._record:
            self.ready_order = []

    def _comm_stream(self, device):
        """The stream the exchange is issued from.  A COMPUTE stream must never wait for it: the model runs its two branches on two HIP
        streams and a bucket holds gradients of both, so issuing the reduction from whichever stream delivered the bucket's last
        gradient (round 2) made that branch's backward wait for the other branch's -- measured with a one-rank RCCL communicator on
        MI355X: 27.6 -> 37.1 ms per step.  Only this stream waits."""
        key = (device.type, device.index)
        st = self._comm.get(key)
        if st is None:
            st = self._comm[key] = torch.cuda.Stream(device=device)
        return st

    def _pack(self, b, comm=None):
        """The bucket's gradients, as autograd left them in p.grad, copied into the flat buffer by one multi-tensor launch;
        afterwards p.grad IS the view (the optimizer reads the reduced values there).  Round 2 kept p.grad as views all the time,
        which turns every AccumulateGrad into a read-modify-write kernel of its own: ~330 extra launches per step."""
        src, dst = [], []
        for p, v in zip(b.params, b.views):
            if p.grad is None:
                v.zero_()            # no gradient this step: contributes 0 to the average (as DDP does)
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
            if comm is not None:
                for g in src:
                    g.record_stream(comm)    # produced on a branch stream, read here: keep the allocator from recycling it early
        for p, v in zip(b.params, b.views):
            p.grad = v

    def _launch(self, b):
        b.launched = True
        if not self.active:
            self._pack(b)
            return
        if b.flat.is_cuda:
            comm = self._comm_stream(b.flat.device)
            # order the reduction after everything queued so far on every stream that accumulated one of the bucket's gradients
            # (one event per stream, recorded now: later than strictly needed, but no event per gradient and nothing blocks compute)
            for st in b.streams.values():
                ev = torch.cuda.Event()
                ev.record(st)
                comm.wait_event(ev)
            with torch.cuda.stream(comm):
                self._pack(b, comm)
                if self.world > 1:
                    b.flat.div_(self.world)
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            self._pack(b)
            if self.world > 1:
                b.flat.div_(self.world)
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _launch_ready_prefix(self):
        while self.next_to_launch < len(self.buckets) and self.buckets[self.next_to_launch].pending == 0:
            self._launch(self.buckets[self.next_to_launch])
            self.next_to_launch += 1

    def _on_grad_ready(self, p):
        if self._record:
            self.ready_order.append(p)
        b = self.buckets[self.bucket_of[p]]
        b.pending -= 1
        if p.is_cuda and self.active:
            st = torch.cuda.current_stream()     # the stream that just accumulated this gradient
            b.streams[st.cuda_stream] = st
        if self._rebuilt:  # overlap only once the bucket order follows the backward
            self._launch_ready_prefix()

    def finish(self):
        """Call after backward: launches what has not gone out yet (in bucket order) and waits."""
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()     # NCCL: the CURRENT stream waits for the collective (no host block); gloo: the host waits
                b.work = None
        self.step_idx += 1

    def allreduce_ms(self, reps: int = 3):
        """Standalone cost of one step's gradient exchange: every bucket all-reduced back to back with nothing else on the
        device, HIP-event time per repetition.  Overwrites the gradients (call it outside a step); None when no collective
        would be issued."""
        if not self.active:
            return None
        flat = [b.flat for b in self.buckets]
        if not flat or not flat[0].is_cuda:
            return None
        for t in flat:                      # one untimed pass: communicator / channel set-up
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True) for t in flat]
            for w in works:
                w.wait()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
