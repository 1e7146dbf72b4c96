"""Inference (eval mode, no grad) frames/s of the fusion forward at batch 1 and 4, alternating resident batches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd import gemm_tuning

gemm_tuning.enable(0)
cfg = fusion_cfg("middle")
torch.manual_seed(0)
model, _, _ = build_model(cfg)
model = model.cuda().eval()
for batch in (1, 4):
    datas = [build_inputs(cfg, batch, "kitti", 0, torch.device("cuda"), cycle=c)[1] for c in range(2)]
    with torch.no_grad():
        for i in range(6):
            model(datas[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        for i in range(n):
            model(datas[i % 2])
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print("eval batch %d: %.2f ms/step, %.1f frames/s" % (batch, ms, batch * 1e3 / ms))
