"""Soak: N training steps over alternating batches; device memory and step time at intervals (leaks, drift).
usage: python tools/probes/soak.py [steps=400] [batch=4] [--reducer]      (the index prefetch of bench.py is on: step(batch, next_batch);
--reducer: a one-rank RCCL communicator and the GradReducer with forced collectives, i.e. the N > 1 step on one GPU)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.trainer import TrainStep
from fusiontransformer_amd import gemm_tuning

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(args[0]) if len(args) > 0 else 400
batch = int(args[1]) if len(args) > 1 else 4
gemm_tuning.enable(0)
cfg = fusion_cfg("middle")
torch.manual_seed(0)
model, m2d, m3d = build_model(cfg)
model = model.cuda().train()
reducer = None
if "--reducer" in sys.argv:
    from fusiontransformer_amd.dist import GradReducer, init_process_group
    init_process_group("nccl", force=True)
    reducer = GradReducer(model, force_collectives=True)
step = TrainStep(cfg, model, metrics=(m2d, m3d), grad_reducer=reducer)
datas = [build_inputs(cfg, batch, "kitti", 0, torch.device("cuda"), cycle=c)[1] for c in range(3)]
for i in range(6):
    step(datas[i % 3], datas[(i + 1) % 3])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(datas[i % 3], datas[(i + 1) % 3])
    if (i + 1) % 100 == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        l = step.last
        print("step %4d  %.2f ms/step  allocated %.0f MiB  reserved %.0f MiB  loss_2d %.4f loss_3d %.4f" % (
            i + 1, (t1 - t0) * 10, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, float(l["loss_2d"]), float(l["loss_3d"])) + "  prefetch pause %d" % step._prefetch_pause, flush=True)
        t0 = time.perf_counter()
if reducer is not None:
    import torch.distributed as dist
    dist.destroy_process_group()
