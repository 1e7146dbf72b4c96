"""In-process A/B of a switch that is read at call time: alternates the two settings in blocks of steps on the SAME model / batches / box,
several rounds, and prints the median step time of each (a step = TrainStep on alternating resident batches, synchronised at block ends).
usage: python tools/probes/ab_env_toggle.py VAR A_VALUE B_VALUE [batch] [rounds] [block]"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.trainer import TrainStep

var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 4
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 6
block = int(sys.argv[6]) if len(sys.argv) > 6 else 20
from fusiontransformer_amd import gemm_tuning
gemm_tuning.enable(0)
cfg = fusion_cfg("middle")
torch.manual_seed(0)
model, m2d, m3d = build_model(cfg)
model = model.cuda().train()
step = TrainStep(cfg, model, metrics=(m2d, m3d))
datas = [build_inputs(cfg, batch, "kitti", 0, torch.device("cuda"), cycle=c)[1] for c in range(2)]
for v in (va, vb):
    os.environ[var] = v
    for i in range(6):
        step(datas[i % 2])
torch.cuda.synchronize()
res = {va: [], vb: []}
i = 0
for r in range(rounds):
    for v in ((va, vb) if r % 2 == 0 else (vb, va)):
        os.environ[var] = v
        for _ in range(3):
            step(datas[i % 2]); i += 1
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(block):
            step(datas[i % 2]); i += 1
        torch.cuda.synchronize()
        res[v].append(1e3 * (time.perf_counter() - t) / block)
for v in (va, vb):
    print("%s=%s batch %d: median %.2f ms/step over %d blocks of %d steps (min %.2f, max %.2f)" % (var, v, batch, statistics.median(res[v]), len(res[v]), block, min(res[v]), max(res[v])))
