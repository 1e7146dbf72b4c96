"""Time of the fused loss call (3 launches) at the bench's point counts.  usage: python tools/probes/loss_time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusiontransformer_amd import functional as spf
for n in (22681, 81237):
    torch.manual_seed(0)
    preds = {k: torch.randn(n, 20, device="cuda") for k in ("lidar_seg_logit", "img_seg_logit", "lidar_seg_logit2", "img_seg_logit2")}
    label = torch.randint(0, 20, (n,), device="cuda")
    cw = torch.rand(20, device="cuda") + 0.5
    for dual, lam in ((True, 0.1), (False, 0.1), (True, 0.0)):
        for _ in range(5):
            spf.fusion_loss(preds, label, cw, lam, dual)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            spf.fusion_loss(preds, label, cw, lam, dual)
        e1.record()
        torch.cuda.synchronize()
        print("n=%6d dual=%s lambda=%.1f: %.1f us per call (3 launches, back to back)" % (n, dual, lam, e0.elapsed_time(e1) * 1e3 / 50))
