"""Data parallel rehearsal on ONE GPU: two fresh processes share cuda:0 and exchange gradients over gloo, with the real fusion
model (depth-2 trunk), its two branch streams and the graphed trunk segments all on -- the N>1 code path of bench.py
(fusiontransformer_amd/dist.py: bucketed all-reduce issued from post-accumulate hooks) with everything but RCCL itself.

Checked per step: the gradients the reducer leaves in p.grad equal the mean of the ranks' local gradients (computed by a twin
model without the reducer on the same batch, same dropout masks); after three optimizer steps every parameter is bit-identical
across ranks (BatchNorm statistics stay per replica, as in the reference: modules/TorchpackInterface.py:44-81 has no SyncBN)."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        from fusiontransformer_amd.data.synth import make_batch
        from fusiontransformer_amd.dist import GradReducer, init_process_group
        from fusiontransformer_amd.models.build import build_model
        from fusiontransformer_amd.trainer import TrainStep, fusion_losses
        from tests.helpers import product_inputs, small_cfg
        init_process_group("gloo")
        cfg = small_cfg("middle")
        torch.manual_seed(100 + rank)          # different initial weights per rank: the reducer broadcasts rank 0's
        model, _, _ = build_model(cfg)
        twin, _, _ = build_model(cfg)
        model, twin = model.cuda().train(), twin.cuda().train()
        assert model.image_backbone.backbone.graph_taps, "graphed trunk segments are expected to be on"
        red = GradReducer(model, bucket_mb=8.0)    # several buckets at depth 2
        step = TrainStep(cfg, model, grad_reducer=red)
        cw = torch.tensor(cfg.TRAIN.CLASS_WEIGHTS, device="cuda")
        worst = 0.0
        for s in range(3):
            pin = product_inputs(make_batch([20 * s + 2 * rank, 20 * s + 2 * rank + 1], max_points=1500 + 100 * rank))
            # this rank's purely local gradients: a twin with the same parameters, no reducer, same dropout masks
            twin.load_state_dict(model.state_dict())
            twin.zero_grad(set_to_none=True)
            torch.manual_seed(7 + s)
            out = twin(pin)
            l2, l3 = fusion_losses(out, pin["seg_label"], cw, 0.1, True)
            (l2 + l3).backward()
            local = {n: p.grad.detach().clone() for n, p in twin.named_parameters() if p.grad is not None}
            step.fused_loss = False            # same loss code path as the twin
            torch.manual_seed(7 + s)
            step(pin)                          # begin_step, forward (2 streams, graphs), backward with overlapped all-reduces, finish, Adam
            torch.cuda.synchronize()
            for n, p in model.named_parameters():
                if not p.requires_grad:
                    continue
                assert n in local, "trainable parameter %s received no local gradient" % n
                g = local[n]
                parts = [torch.zeros_like(g) for _ in range(world)]
                dist.all_gather(parts, g)
                ref = sum(parts) / world
                denom = max(ref.abs().max().item(), 1e-6)
                worst = max(worst, (p.grad - ref).abs().max().item() / denom)
        # parameters after three steps: identical on every rank
        sig = torch.stack([torch.stack([p.detach().double().sum(), p.detach().double().abs().sum()]) for p in model.parameters()]).cpu()
        sigs = [torch.zeros_like(sig) for _ in range(world)]
        dist.all_gather(sigs, sig)
        same = all(torch.equal(sigs[0], t) for t in sigs)
        captured = bool(model.image_backbone.backbone.__dict__.get("_graph_cache")) and \
            all(v is not None for v in model.image_backbone.backbone._graph_cache.values())
        q.put((rank, "ok", worst, same, red._rebuilt, len(red.buckets), captured))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as err:   # report instead of hanging the parent on q.get
        import traceback
        q.put((rank, "error", traceback.format_exc(), False, False, 0, False))
        raise


def test_two_ranks_on_one_gpu_real_model_streams_and_graphs():
    import multiprocessing as mp
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, status, worst, same, rebuilt, nb, captured in res:
        assert status == "ok", "rank %d failed:\n%s" % (rank, worst)
        # (g0/2 + g1/2) vs (g0 + g1)/2 in float32: rounding only
        assert worst < 1e-5, (rank, worst)
        assert same, "parameters differ across ranks after 3 steps"
        assert rebuilt and nb > 1, (rebuilt, nb)
        assert captured, "the trunk was not running as HIP graphs"
    for p in procs:
        assert p.exitcode == 0


def _rccl_one_rank_worker(port, q):
    """The collective path of the N > 1 step on ONE GPU over the real backend: a one-rank RCCL communicator, `async_op` all-reduces
    issued from the post-accumulate hooks while the two branch streams are still running the backward, `work.wait()` in finish().
    An all-reduce over one rank is the identity, so every gradient and, after three Adam steps, every parameter must equal a twin's
    that runs without any reducer -- bit for bit (reference wiring: modules/TorchpackInterface.py:44-81)."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        from fusiontransformer_amd.data.synth import make_batch
        from fusiontransformer_amd.dist import GradReducer, init_process_group
        from fusiontransformer_amd.models.build import build_model
        from fusiontransformer_amd.trainer import TrainStep
        from tests.helpers import product_inputs, small_cfg
        init_process_group("nccl", force=True)
        assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
        cfg = small_cfg("middle")
        torch.manual_seed(5)
        model, _, _ = build_model(cfg)
        twin, _, _ = build_model(cfg)
        twin.load_state_dict(model.state_dict())
        model, twin = model.cuda().train(), twin.cuda().train()
        red = GradReducer(model, bucket_mb=8.0, force_collectives=True)
        assert red.active
        step, step_twin = TrainStep(cfg, model, grad_reducer=red), TrainStep(cfg, twin)
        same_grads = True
        for s in range(3):
            pin = product_inputs(make_batch([30 + 2 * s, 31 + 2 * s], max_points=1800))
            torch.manual_seed(9 + s)
            step(pin)
            torch.manual_seed(9 + s)
            step_twin(pin)
            torch.cuda.synchronize()
            for (n, p), (_, t) in zip(model.named_parameters(), twin.named_parameters()):
                if p.requires_grad:
                    same_grads = same_grads and t.grad is not None and torch.equal(p.grad, t.grad)
        same_params = all(torch.equal(p, t) for p, t in zip(model.parameters(), twin.parameters()))
        ms = red.allreduce_ms(2)
        captured = model.image_backbone.backbone.graph_state()
        q.put(("ok", same_grads, same_params, red._rebuilt, len(red.buckets), ms, captured))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(("error", traceback.format_exc(), False, False, 0, None, ""))
        raise


def test_one_rank_rccl_collective_path_is_the_identity():
    import multiprocessing as mp
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(_free_port(), q))
    p.start()
    status, same_grads, same_params, rebuilt, nb, ms, captured = q.get(timeout=600)
    p.join(timeout=120)
    assert status == "ok", same_grads
    assert same_grads, "gradients after the one-rank RCCL all-reduce differ from the local ones"
    assert same_params, "parameters after three steps differ from the twin without a reducer"
    assert rebuilt and nb > 1 and ms is not None and ms > 0
    assert captured == "on", captured
    assert p.exitcode == 0
