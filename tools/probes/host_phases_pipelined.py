"""Host time of the phases of the training step WITHOUT a synchronise per step (the way bench.py runs it): forward issue (host reads
accounted separately), loss, backward issue, optimizer.  usage: python tools/probes/host_phases_pipelined.py [batch] [steps]"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.trainer import TrainStep
from fusiontransformer_amd import functional as spf, gemm_tuning, sparse

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
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
waited = [0.0]
def values(self):
    t = time.perf_counter(); self.event.synchronize(); waited[0] += time.perf_counter() - t
    return [int(v) for v in self.host.tolist()]
sparse.HostRead.values = values
acc = {}
def mark(name, t0, sub=0.0):
    t = time.perf_counter(); acc.setdefault(name, []).append((t - t0 - sub) * 1e3); return t
for sync_each in (False, True):
    acc.clear()
    t_all = time.perf_counter()
    for it in range(steps):
        data = datas[it % 2]
        t = time.perf_counter()
        step.optimizer.zero_grad(set_to_none=True); t = mark("zero_grad", t)
        waited[0] = 0.0
        preds = model(data); w = waited[0]; acc.setdefault("host reads (blocked)", []).append(w * 1e3); t = mark("forward issue", t, w)
        conf = {"3d": None, "2d": None}
        for m in step.metrics:
            conf["3d" if "3d" in m.name else "2d"] = m.mat
        l2, l3 = spf.fusion_loss(preds, data["seg_label"], step.class_weights, step.lambda_xm, step.dual_head, conf3d=conf["3d"], conf2d=conf["2d"]); t = mark("loss issue", t)
        (l2 + l3).backward(); t = mark("backward issue", t)
        step.optimizer.step(); t = mark("optimizer issue", t)
        if sync_each:
            torch.cuda.synchronize(); t = mark("GPU tail", t)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) / steps * 1e3
    print("batch %d, %s: %.2f ms/step" % (batch, "synchronise after every step" if sync_each else "pipelined", wall))
    for k, v in acc.items():
        print("   %-22s %7.2f ms (median)" % (k, statistics.median(v)))
