"""In-process A/B of the two-stream branch overlap (model.overlap_branches) at a given batch: alternating blocks of steps on the same
model / batches / box.  usage: python tools/probes/ab_overlap.py [batch] [rounds] [block]"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.trainer import TrainStep
from fusiontransformer_amd import gemm_tuning

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
block = int(sys.argv[3]) if len(sys.argv) > 3 else 30
gemm_tuning.enable(0)
cfg = fusion_cfg("middle")
torch.manual_seed(0)
model, m2d, m3d = build_model(cfg)
model = model.cuda().train()
step = TrainStep(cfg, model, metrics=(m2d, m3d))
datas = [build_inputs(cfg, batch, "kitti", 0, torch.device("cuda"), cycle=c)[1] for c in range(2)]
for v in (True, False):
    model.overlap_branches = v
    for i in range(6):
        step(datas[i % 2])
torch.cuda.synchronize()
res = {True: [], False: []}
i = 0
for r in range(rounds):
    for v in ((True, False) if r % 2 == 0 else (False, True)):
        model.overlap_branches = v
        for _ in range(3):
            step(datas[i % 2]); i += 1
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(block):
            step(datas[i % 2]); i += 1
        torch.cuda.synchronize()
        res[v].append(1e3 * (time.perf_counter() - t) / block)
for v in (True, False):
    print("overlap_branches=%s batch %d: median %.2f ms/step over %d blocks of %d steps (min %.2f, max %.2f)" % (v, batch, statistics.median(res[v]), len(res[v]), block, min(res[v]), max(res[v])))
