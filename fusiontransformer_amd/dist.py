"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

Counterpart of the reference's DDP wiring (modules/TorchpackInterface.py:44-81:
`dist.init()` + `DistributedDataParallel(find_unused_parameters=True)`).  Frames
are independent units (the batch index is the 4th coordinate column,
data/collate.py:41-42), so ranks share nothing in the forward; the only exchange
is the gradient average, ~108 M fp32 = 432 MB per step.

Design for xGMI (point-to-point links, no switch): few large buckets (default
128 MB: four per step) so each all-reduce amortises its launch, RCCL can spread
it over all 7 links, and the host work of issuing a bucket -- which runs on the
autograd thread, between two backward nodes -- happens three times per backward,
not seven; each bucket is one flat buffer: when its last gradient has arrived the
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
import time

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
    __slots__ = ("params", "flat", "views", "work", "launched", "tail_seen", "hold", "events")

    def __init__(self, params, flat, views):
        self.params, self.flat, self.views = params, flat, views
        self.work, self.launched, self.tail_seen, self.hold, self.events = None, False, False, None, None


class GradReducer:
    """Bucketed, overlapped gradient all-reduce (average) for `model`'s trainable parameters.

    Step 0 runs with a hook on every parameter and records the order in which gradients become ready (and the streams they are
    accumulated on); nothing overlaps yet.  From step 1 on the buckets follow that order (rank 0's, broadcast) and only the LAST
    parameter of each bucket keeps a hook: ~330 Python hook calls per backward cost 2.5-3 ms per step on MI355X (measured with a
    one-rank RCCL communicator: 27.6 ms without a reducer, 30.2-31.6 ms with per-parameter hooks, the collectives themselves free).
    When a bucket's tail arrives, every gradient of the bucket is checked for presence (p.grad was set to None at the start of the
    step); a bucket that is not complete -- the order changed -- simply waits for a later tail or for finish()."""

    def __init__(self, model, bucket_mb: float = 128.0, process_group=None, broadcast_params=True, force_collectives=False):
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
        self._streams = {}          # every stream a gradient was accumulated on (recorded in step 0)
        self.ready_order = []
        self._rebuilt = False
        self._record = False
        self._handles = []
        self.hook_stats = {"calls": 0, "early_launches": 0, "deferred": 0, "host_ms": 0.0}     # host time spent in the bucket-tail hooks (autograd thread)
        if self.active and broadcast_params:
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0, group=self.pg)
        self._build(list(reversed(self.params)))
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad_ready) for p in self.params]
        # The hooks keep the AccumulateGrad nodes (created on the default stream) alive while the model's
        # branches run on their own streams; autograd then orders each accumulation after its producer
        # stream, which is exactly what the exchange relies on.  The advisory warning about it is noise.
        try:
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        except AttributeError:
            pass

    # -- bucket construction -------------------------------------------------
    def _build(self, ordered):
        self.buckets = []
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
            self.buckets.append(_Bucket(g, flat, views))
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
            for h in self._handles:
                h.remove()
            # one hook per bucket, on the parameter whose gradient arrived last in step 0
            self._handles = [b.params[-1].register_post_accumulate_grad_hook(lambda p, i=i: self._on_bucket_tail(i)) for i, b in enumerate(self.buckets)]
        for b in self.buckets:
            b.work, b.launched, b.tail_seen, b.hold, b.events = None, False, False, None, None
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

    def _pack(self, b, early=False):
        """The bucket's gradients, as autograd left them in p.grad, copied into the flat buffer by one multi-tensor launch; p.grad
        become the views in finish() (the optimizer reads the reduced values there).  Round 2 kept p.grad as views all the time,
        which turns every AccumulateGrad into a read-modify-write kernel of its own: ~330 extra launches per step.
        early=True (called from an autograd hook): returns False, with nothing done, when a gradient of the bucket is still missing."""
        grads = [p.grad for p in b.params]
        if early and any(g is None for g in grads):
            return False
        src, dst = [], []
        for g, v in zip(grads, b.views):
            if g is None:
                v.zero_()            # no gradient this step: contributes 0 to the average (as DDP does)
            else:
                src.append(g)
                dst.append(v)
        if src and os.environ.get("FTX_REDUCER_NOPACK") != "1":     # measurement aid: 1 = host work only, no copy issued
            torch._foreach_copy_(dst, src)
            # produced on a branch stream, read on the exchange stream: kept alive until finish() has made the step's stream wait for the
            # exchange (tensor.record_stream would do, at an event per gradient when it is freed)
            b.hold = src
        return True

    def _mark(self, b):
        """One event per accumulating stream, recorded NOW: everything that produced this bucket's gradients has been issued."""
        b.events = []
        for st in self._streams.values():
            ev = torch.cuda.Event()
            ev.record(st)
            b.events.append(ev)

    def _launch(self, b, early=False):
        """Everything here runs on the autograd thread when early=True, between two backward nodes: host time spent here delays the
        issue of the rest of the backward one to one (measured: 7 early launches of ~0.4 ms each cost MORE than not overlapping the
        exchange at all on one GPU), hence few, large buckets and nothing in this path that can wait for finish()."""
        if not self.active:
            b.launched = self._pack(b, early)
            return b.launched
        if b.flat.is_cuda:
            comm = self._comm_stream(b.flat.device)
            with torch.cuda.stream(comm):
                # order the exchange after everything that had been queued on every gradient-accumulating stream when the bucket's
                # tail arrived (_mark: one event per stream, nothing blocks compute)
                if b.events is None:
                    self._mark(b)
                for ev in b.events:
                    comm.wait_event(ev)
                if not self._pack(b, early):
                    return False
                if self.world > 1:
                    b.flat.div_(self.world)
                if os.environ.get("FTX_REDUCER_NOCOMM") != "1":      # measurement aid: everything but the collective itself
                    b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            if not self._pack(b, early):
                return False
            if self.world > 1:
                b.flat.div_(self.world)
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        b.launched = True
        return True

    def _launch_ready_prefix(self, upto):
        """Buckets go out strictly in bucket order on every rank (collectives are matched by issue order); `upto`: first bucket NOT to
        launch now."""
        while self.next_to_launch < min(upto, len(self.buckets)):
            b = self.buckets[self.next_to_launch]
            if not b.tail_seen or not self._launch(b, early=True):
                return
            self.next_to_launch += 1

    def _on_grad_ready(self, p):
        """Step 0 (a hook on every parameter): learn the ready order and the accumulating streams; nothing is launched early."""
        if self._record:
            self.ready_order.append(p)
        if p.is_cuda:
            st = torch.cuda.current_stream()     # the stream that just accumulated this gradient
            self._streams[st.cuda_stream] = st

    def _on_bucket_tail(self, i):
        """Bucket i is complete on the host side: mark it (events on the accumulating streams) and launch the buckets BEFORE it.
        One bucket of delay on purpose.  The host issues the backward several milliseconds ahead of the GPU, so a stream-side wait
        enqueued at a bucket's own tail stays pending for that long, and pending waits are expensive on this stack: +2.5 ms per step
        on one GPU with no byte moved, all of it gone when the waits are removed (why is not fully established: the exchange stream
        shares the image branch's hardware queue -- four queues for six streams --, yet the exchange on the default stream costs the
        same).  Waiting for the PREVIOUS bucket's events, which are (nearly) reached by the time the next
        tail arrives, costs +0.5-0.8 ms (tools/probes/reducer_ab6.sh)."""
        t0 = time.perf_counter()
        b = self.buckets[i]
        b.tail_seen = True
        if b.flat.is_cuda and self.active:
            self._mark(b)
        if os.environ.get("FTX_REDUCER_LATE") != "1":       # measurement aid: 1 = nothing goes out before finish()
            before = self.next_to_launch
            self._launch_ready_prefix(i if os.environ.get("FTX_REDUCER_EAGER") != "1" else i + 1)   # aid: 1 = round 3's launch at the tail
            self.hook_stats["early_launches"] += self.next_to_launch - before
            self.hook_stats["deferred"] += int(self.next_to_launch == before)
        self.hook_stats["calls"] += 1
        self.hook_stats["host_ms"] += 1e3 * (time.perf_counter() - t0)

    def finish(self):
        """Call after backward: launches what has not gone out yet (in bucket order) and waits."""
        if self.buckets and self.buckets[0].flat.is_cuda:
            cur = torch.cuda.current_stream()
            self._streams[cur.cuda_stream] = cur
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()     # NCCL: the CURRENT stream waits for the collective (no host block); gloo: the host waits
                b.work = None
        if self.active and self.buckets and self.buckets[0].flat.is_cuda:
            # packs without a collective behind them (FTX_REDUCER_NOCOMM) and the held gradients: the step's stream waits for the exchange stream
            torch.cuda.current_stream().wait_stream(self._comm_stream(self.buckets[0].flat.device))
        for b in self.buckets:
            b.hold = None
            for p, v in zip(b.params, b.views):
                p.grad = v
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
