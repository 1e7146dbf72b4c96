"""Host issue time vs GPU wall time of a training step (is the host feeding both streams fast enough?).
usage: python tools/host_vs_gpu.py [batch]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.trainer import TrainStep

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = fusion_cfg("middle")
torch.manual_seed(0)
model, m2d, m3d = build_model(cfg)
model = model.cuda().train()
step = TrainStep(cfg, model, metrics=(m2d, m3d))
_, data = build_inputs(cfg, batch, "kitti", 0, torch.device("cuda"))
for _ in range(6):
    step(data)
torch.cuda.synchronize()
for overlap in (True, False):
    model.overlap_branches = overlap
    for _ in range(2):
        step(data)
    torch.cuda.synchronize()
    host, wall = [], []
    for _ in range(8):
        t0 = time.perf_counter()
        step(data)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
    host.sort(); wall.sort()
    print("batch %d overlap=%s: host issue %.2f ms, step wall (with sync) %.2f ms, GPU tail after last launch %.2f ms" % (
        batch, overlap, host[len(host) // 2], wall[len(wall) // 2], wall[len(wall) // 2] - host[len(host) // 2]))
