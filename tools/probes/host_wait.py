"""Is the training step host-bound or GPU-bound at a given batch?  Runs blocks of steps with one synchronise at the block's end and
accounts the host thread's time: blocked in the step's host reads (sparse.HostRead), blocked in the final synchronise, busy otherwise
(issuing).  Busy ~ wall => the host thread is the limit; large waits => the GPU is.  usage: python tools/probes/host_wait.py [batch] [block]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.trainer import TrainStep
from fusiontransformer_amd import gemm_tuning, sparse

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
block = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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

waited = [0.0, 0]
orig = sparse.HostRead.values
def values(self):
    t = time.perf_counter()
    self.event.synchronize()
    waited[0] += time.perf_counter() - t
    waited[1] += 1
    return [int(v) for v in self.host.tolist()]
sparse.HostRead.values = values

for rep in range(3):
    waited[0], waited[1] = 0.0, 0
    t0 = time.perf_counter()
    for i in range(block):
        step(datas[i % 2])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    wall = (t2 - t0) / block * 1e3
    print("batch %d: %.2f ms/step wall; host blocked in %d reads/step %.2f ms, in the final synchronise %.2f ms/step, issuing %.2f ms/step"
          % (batch, wall, waited[1] // block, waited[0] / block * 1e3, (t2 - t1) / block * 1e3, (t1 - t0 - waited[0]) / block * 1e3))
