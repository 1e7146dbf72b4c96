"""Adam stepped by one libftx launch (csrc/ftx_optim.hip).

`fusiontransformer_amd.optim.Adam` is `torch.optim.Adam` -- same constructor, same update rule (L2 weight decay, no amsgrad), same
`state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so checkpoints move both ways -- with `step()` replaced:
the reference's `optimizer.step()` (common/solver/build.py:7-20 builds it, modules/SemanticTrainer.py steps it once per batch) walks
~300 parameter tensors; torch's fused implementation does that in 14 multi-tensor launches and ~0.9 ms of host time, this one in one
launch whose per-step host work is one pass over the gradients' addresses.

Anything the kernel does not cover (CPU parameters, non-float32, sparse gradients, amsgrad / maximize / capturable / differentiable)
is handed to torch.optim.Adam.step unchanged."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

_REC = np.dtype([("p", "<i8"), ("m", "<i8"), ("v", "<i8"), ("g", "<i8"), ("n", "<i8"), ("step_size", "<f4"), ("inv_bc2_sqrt", "<f4")])


class _GroupPlan:
    """Static description of one parameter group for the kernel: the table rows and the chunk -> (tensor, offset) map."""

    def __init__(self, params, states, device):
        L = _lib.load()
        assert int(L.ftx_adam_tensor_bytes()) == _REC.itemsize
        chunk = int(L.ftx_adam_chunk_elements())
        self.params = params
        self.steps = np.array([int(float(st["step"])) for st in states], dtype=np.int64)
        n = np.array([p.numel() for p in params], dtype=np.int64)
        per = (n + chunk - 1) // chunk
        self.n_chunks = int(per.sum())
        tid = np.repeat(np.arange(len(params), dtype=np.int32), per)
        off = np.concatenate([np.arange(k, dtype=np.int64) * chunk for k in per]) if self.n_chunks else np.zeros(0, np.int64)
        self.chunk_tensor = torch.from_numpy(tid).to(device)
        self.chunk_offset = torch.from_numpy(off).to(device)
        self.host = [torch.empty(len(params) * _REC.itemsize, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        self.rec = [h.numpy().view(_REC) for h in self.host]
        self.events = [None, None]
        self.turn = 0
        self.table = torch.empty(len(params) * _REC.itemsize, dtype=torch.uint8, device=device)
        for r in self.rec:
            r["m"] = [st["exp_avg"].data_ptr() for st in states]
            r["v"] = [st["exp_avg_sq"].data_ptr() for st in states]
            r["n"] = n
        self.keep = states     # the moments the table points at


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, **kwargs):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kwargs)
        self._plans = {}

    # ---- state_dict compatibility: the per-parameter step counters live in numpy between steps
    def _sync_steps(self):
        for plan in self._plans.values():
            for p, t in zip(plan.params, plan.steps):
                self.state[p]["step"] = torch.tensor(float(t))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}      # moments were replaced: rebuild the tables

    def _eligible(self, group):
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
            return False
        if isinstance(group["lr"], torch.Tensor):
            return False
        dev = group["params"][0].device if group["params"] else None
        for p in group["params"]:
            if not (p.is_cuda and p.device == dev and p.dtype == torch.float32 and p.is_contiguous()):
                return False
            if p.grad is not None and (p.grad.is_sparse or p.grad.dtype != torch.float32):
                return False
        return len(group["params"]) > 0

    def _plan(self, gi, group):
        plan = self._plans.get(gi)
        params = group["params"]
        if plan is not None and len(plan.params) == len(params) and all(a is b for a, b in zip(plan.params, params)):
            return plan
        states = []
        for p in params:
            st = self.state[p]
            if len(st) == 0:     # what torch.optim.Adam._init_group creates lazily
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            states.append(st)
        plan = self._plans[gi] = _GroupPlan(list(params), states, params[0].device)
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        if not all(self._eligible(g) for g in self.param_groups):
            self._sync_steps()
            self._plans = {}
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.load()
        for gi, group in enumerate(self.param_groups):
            plan = self._plan(gi, group)
            beta1, beta2 = group["betas"]
            grads = [p.grad for p in plan.params]
            keep = []
            gptr = np.zeros(len(grads), dtype=np.int64)
            for i, g in enumerate(grads):
                if g is not None:
                    if not g.is_contiguous():
                        g = g.contiguous()
                        keep.append(g)
                    gptr[i] = g.data_ptr()
            live = gptr != 0
            plan.steps[live] += 1
            t = np.maximum(plan.steps, 1).astype(np.float64)
            k = plan.turn
            plan.turn ^= 1
            if plan.events[k] is not None:
                plan.events[k].synchronize()          # the copy that last read this pinned buffer has been issued AND has run
            rec = plan.rec[k]
            rec["p"] = [p.data_ptr() for p in plan.params]
            rec["g"] = gptr
            rec["step_size"] = (float(group["lr"]) / (1.0 - beta1 ** t)).astype(np.float32)
            rec["inv_bc2_sqrt"] = (1.0 / np.sqrt(1.0 - beta2 ** t)).astype(np.float32)
            with torch.cuda.device(plan.table.device):
                plan.table.copy_(plan.host[k], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                plan.events[k] = ev
                _lib.check(L.ftx_adam_step(plan.table.data_ptr(), plan.chunk_tensor.data_ptr(), plan.chunk_offset.data_ptr(), plan.n_chunks,
                                           float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]), _lib.stream()),
                           "ftx_adam_step")
            del keep
        return loss
