"""GPU-side spans of the phases of a pipelined training step (no synchronise per step): timing events recorded on the step's own stream
after the forward, the loss, the backward and the optimizer -- each of those points is a join of the two branch streams, so the spans
are the critical path of the step on the GPU's clock.  usage: python tools/probes/phase_spans.py [batch] [steps]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.trainer import TrainStep
from fusiontransformer_amd import functional as spf, gemm_tuning

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
gemm_tuning.enable(0)
cfg = fusion_cfg("middle")
torch.manual_seed(0)
model, m2d, m3d = build_model(cfg)
model = model.cuda().train()
step = TrainStep(cfg, model, metrics=(m2d, m3d))
datas = [build_inputs(cfg, batch, "kitti", 0, torch.device("cuda"), cycle=c)[1] for c in range(2)]
for i in range(8):
    step(datas[i % 2])
torch.cuda.synchronize()
marks = []
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(steps):
    data = datas[it % 2]
    e0 = ev()
    step.optimizer.zero_grad(set_to_none=True)
    preds = model(data); e1 = ev()
    conf = {"3d": None, "2d": None}
    for m in step.metrics:
        conf["3d" if "3d" in m.name else "2d"] = m.mat
    l2, l3 = spf.fusion_loss(preds, data["seg_label"], step.class_weights, step.lambda_xm, step.dual_head, conf3d=conf["3d"], conf2d=conf["2d"]); e2 = ev()
    (l2 + l3).backward(); e3 = ev()
    step.optimizer.step(); e4 = ev()
    marks.append((e0, e1, e2, e3, e4))
torch.cuda.synchronize()
names = ["forward (both branches, to the join)", "loss", "backward (to the join)", "optimizer"]
tot = 0
for k, n in enumerate(names):
    v = statistics.median(m[k].elapsed_time(m[k + 1]) for m in marks[5:])
    tot += v
    print("%-40s %7.2f ms" % (n, v))
gap = statistics.median(marks[i][4].elapsed_time(marks[i + 1][0]) for i in range(5, steps - 1))
print("%-40s %7.2f ms" % ("between steps", gap))
print("%-40s %7.2f ms per step at batch %d" % ("sum", tot + gap, batch))
