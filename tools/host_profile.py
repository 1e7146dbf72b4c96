"""Where the host time of one training step goes (batch 1: the GPU is never the bottleneck)."""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.trainer import TrainStep

cfg = fusion_cfg("middle")
torch.manual_seed(0)
model, m2d, m3d = build_model(cfg)
model = model.cuda().train()
step = TrainStep(cfg, model, metrics=(m2d, m3d))
_, data = build_inputs(cfg, int(sys.argv[1]) if len(sys.argv) > 1 else 1, "kitti", 0, torch.device("cuda"))
for _ in range(4):
    step(data)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    step(data)
torch.cuda.synchronize()
print("ms/step %.2f" % ((time.perf_counter() - t) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step(data)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
