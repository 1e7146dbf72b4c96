"""Host time of the phases of one training step (forward issue / loss / backward issue / optimizer) at a given batch.
usage: python tools/host_phases.py [batch] [--reducer]     (--reducer: one-rank RCCL communicator + GradReducer, the N > 1 step on one GPU)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.trainer import TrainStep
from fusiontransformer_amd import functional as spf

args = [a for a in sys.argv[1:] if not a.startswith("--")]
batch = int(args[0]) if args else 4
use_reducer = "--reducer" in sys.argv
cfg = fusion_cfg("middle")
torch.manual_seed(0)
model, m2d, m3d = build_model(cfg)
model = model.cuda().train()
reducer = None
if use_reducer:
    from fusiontransformer_amd.dist import GradReducer, init_process_group
    init_process_group("nccl", force=True)
    reducer = GradReducer(model, force_collectives=True)
step = TrainStep(cfg, model, metrics=(m2d, m3d), grad_reducer=reducer)
_, data = build_inputs(cfg, batch, "kitti", 0, torch.device("cuda"))
for _ in range(6):
    step(data)
torch.cuda.synchronize()
acc = {}
def mark(name, t0):
    t = time.perf_counter(); acc.setdefault(name, []).append((t - t0) * 1e3); return t
for it in range(8):
    t = time.perf_counter()
    if reducer is not None:
        reducer.begin_step()
    else:
        step.optimizer.zero_grad(set_to_none=True)
    t = mark("zero_grad", t)
    preds = model(data); t = mark("forward issue", t)
    conf = {"3d": None, "2d": None}
    for m in step.metrics:
        conf["3d" if "3d" in m.name else "2d"] = m.mat
    l2, l3 = spf.fusion_loss(preds, data["seg_label"], step.class_weights, step.lambda_xm, step.dual_head, conf3d=conf["3d"], conf2d=conf["2d"]); t = mark("loss issue", t)
    (l2 + l3).backward(); t = mark("backward issue", t)
    if reducer is not None:
        reducer.finish(); t = mark("reducer finish", t)
    step.optimizer.step(); t = mark("optimizer issue", t)
    torch.cuda.synchronize(); t = mark("GPU tail", t)
tot = 0
for k, v in acc.items():
    v.sort(); print("%-18s %7.2f ms" % (k, v[len(v) // 2])); tot += v[len(v) // 2]
print("%-18s %7.2f ms" % ("sum", tot))
