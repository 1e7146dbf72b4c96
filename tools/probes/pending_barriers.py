"""What does a stream-side wait that stays pending for milliseconds cost the rest of the GPU?  Before every step, K otherwise idle streams
are made to wait for an event recorded at the current tail of the LiDAR branch's stream (the host runs a step ahead of the GPU, so the
event completes several milliseconds later); nothing else is queued on those streams.  Alternating blocks of steps in one process.
usage: python tools/probes/pending_barriers.py [batch] [rounds] [block]"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.models._fusion_common import _branch_streams
from fusiontransformer_amd.trainer import TrainStep
from fusiontransformer_amd import gemm_tuning

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
block = int(sys.argv[3]) if len(sys.argv) > 3 else 20
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
dev = torch.device("cuda", torch.cuda.current_device())
s_img, s_lid = _branch_streams(dev)
extra = [torch.cuda.Stream() for _ in range(4)]
i = 0
def run(n, k):
    global i
    for _ in range(n):
        for st in extra[:k]:
            ev = torch.cuda.Event()
            ev.record(s_lid)
            st.wait_event(ev)
        step(datas[i % 2]); i += 1
KS = (0, 1, 2, 4)
res = {k: [] for k in KS}
for r in range(rounds):
    for k in (KS if r % 2 == 0 else KS[::-1]):
        run(3, k)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(block, k)
        torch.cuda.synchronize()
        res[k].append(1e3 * (time.perf_counter() - t) / block)
for k in KS:
    print("%d pending waits per step on idle streams, batch %d: median %.2f ms/step (min %.2f, max %.2f)" % (k, batch, statistics.median(res[k]), min(res[k]), max(res[k])))
