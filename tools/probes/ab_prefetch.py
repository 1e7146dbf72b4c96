"""In-process A/B of the index prefetch (TrainStep(batch, next_batch)): alternating blocks of steps without it / started only (wait=False) / complete (wait=True) on the same
model / batches / box.  usage: python tools/probes/ab_prefetch.py [batch] [rounds] [block]"""
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
i = 0
MODES = ("none", "start", "full")      # no prefetch / issued up to the first host read (wait=False) / built completely (wait=True)
def run(n, mode):
    global i
    step.prefetch_wait = mode == "full"
    for _ in range(n):
        step(datas[i % 2], datas[(i + 1) % 2] if mode != "none" else None); i += 1
for v in MODES:
    run(6, v)
torch.cuda.synchronize()
res = {m: [] for m in MODES}
for r in range(rounds):
    for v in (MODES if r % 2 == 0 else MODES[::-1]):
        run(3, v)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(block, v)
        torch.cuda.synchronize()
        res[v].append(1e3 * (time.perf_counter() - t) / block)
for v in MODES:
    print("index prefetch %s batch %d: median %.2f ms/step over %d blocks of %d steps (min %.2f, max %.2f)" % (v, batch, statistics.median(res[v]), len(res[v]), block, min(res[v]), max(res[v])))
