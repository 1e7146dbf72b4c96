"""world_size-2 gloo test of the gradient reducer (the N>1 path of bench.py) on the CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from fusiontransformer_amd.dist import GradReducer, init_process_group
    r, w, _ = init_process_group("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)   # different initial weights per rank: the reducer must broadcast rank 0's
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    for p in model[4].parameters():
        pass
    frozen = torch.nn.Linear(4, 4)           # never used: must not break the exchange
    for p in frozen.parameters():
        p.requires_grad_(False)
    model.add_module("frozen", frozen)
    red = GradReducer(model, bucket_mb=0.0005)   # tiny buckets -> several all-reduces, order matters
    w0 = [p.detach().clone() for p in model.parameters()]
    gathered = [torch.zeros_like(w0[0]) for _ in range(world)]
    dist.all_gather(gathered, w0[0])
    assert all(torch.equal(g, gathered[0]) for g in gathered), "parameters not broadcast"
    # a twin without the reducer yields this rank's purely local gradients (once buckets overlap
    # with the backward, p.grad of the reduced model is already averaged when backward returns)
    twin = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    twin.load_state_dict({k: v for k, v in model.state_dict().items() if not k.startswith("frozen")})
    ok = True
    for step in range(4):
        torch.manual_seed(1000 + 10 * step + rank)
        x = torch.randn(5, 8)
        twin.zero_grad()
        twin(x).pow(2).sum().backward()
        local = [p.grad.detach().clone() for p in twin.parameters()]
        red.begin_step()
        loss = model[:5](x).pow(2).sum()
        loss.backward()
        red.finish()
        # reference: average of the per-rank local gradients
        for p, g_local in zip([p for p in model.parameters() if p.requires_grad], local):
            parts = [torch.zeros_like(g_local) for _ in range(world)]
            dist.all_gather(parts, g_local)
            ref = sum(parts) / world
            ok = ok and torch.allclose(p.grad, ref, atol=1e-6)
    q.put((rank, ok, red._rebuilt, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, rebuilt, nb in res:
        assert ok, f"rank {rank}: reduced gradients differ from the average of the local ones"
        assert rebuilt and nb > 1


def _forced_worker(port, q):
    """World size 1 with the collective path forced on: the all-reduce of one rank is the identity, so the reduced gradients must be the
    local ones to the last bit, and every bucket must have gone through a collective."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from fusiontransformer_amd.dist import GradReducer, init_process_group
    with pytest.raises(RuntimeError):
        GradReducer(torch.nn.Linear(2, 2), force_collectives=True)      # no process group yet
    r, w, _ = init_process_group("gloo", force=True)
    assert (r, w) == (0, 1) and dist.is_initialized()
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    twin = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    twin.load_state_dict(model.state_dict())
    red = GradReducer(model, bucket_mb=0.0005, force_collectives=True)
    assert red.active and GradReducer(twin).active is False
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    ok = True
    for step in range(3):
        x = torch.randn(5, 8)
        twin.zero_grad()
        twin(x).pow(2).sum().backward()
        red.begin_step()
        model(x).pow(2).sum().backward()
        red.finish()
        ok = ok and all(torch.equal(p.grad, t.grad) for p, t in zip(model.parameters(), twin.parameters()))
    dist.all_reduce = real
    q.put((ok, len(calls), len(red.buckets), red._rebuilt))
    dist.destroy_process_group()


def test_forced_collectives_at_world_size_one():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    ok, calls, nb, rebuilt = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert ok, "a one-rank all-reduce changed the gradients"
    assert nb > 1 and rebuilt and calls >= 3 * nb, (calls, nb)


def _reorder_worker(port, q):
    """From step 1 only the last-ready parameter of each bucket keeps a hook.  If the gradient-ready ORDER changes in a later step -- the
    tail fires while another gradient of its bucket is still missing -- the bucket must not go out early: it waits for finish()."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from fusiontransformer_amd.dist import GradReducer, init_process_group
    init_process_group("gloo", force=True)
    torch.manual_seed(4)
    a, b = torch.nn.Linear(6, 6), torch.nn.Linear(6, 6)
    model = torch.nn.ModuleList([a, b])
    red = GradReducer(model, bucket_mb=1e-9, force_collectives=True)      # one parameter per bucket
    x = torch.randn(3, 6)

    def step(order):
        red.begin_step()
        outs = {"a": a(x).pow(2).sum(), "b": b(x).pow(2).sum()}
        for name in order:                       # two separate backward calls fix the order in which the gradients arrive
            outs[name].backward()
        red.finish()
        return [p.grad.clone() for p in model.parameters()]

    g0 = step("ab")
    g1 = step("ab")                              # buckets rebuilt in the recorded order, one hook per bucket
    early = red.hook_stats["early_launches"]
    g2 = step("ba")                              # the order flips: nothing may go out before its bucket is complete
    ok = all(torch.equal(u, v) for u, v in zip(g0, g1)) and all(torch.equal(u, v) for u, v in zip(g0, g2))
    q.put((ok, red._rebuilt, len(red.buckets), early, red.hook_stats["deferred"]))
    dist.destroy_process_group()


def test_bucket_waits_when_the_gradient_order_changes():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_reorder_worker, args=(_free_port(), q))
    p.start()
    ok, rebuilt, nb, early, deferred = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert ok, "gradients changed when the ready order changed"
    assert rebuilt and nb == 4 and early >= 1 and deferred >= 1, (nb, early, deferred)
