"""Host issue time of the LiDAR branch alone, stage by stage (forward), and of its backward, against the GPU time of the same work.
usage: python tools/probes/lidar_host_stages.py [batch]"""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.sparse import SparseTensor

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = fusion_cfg("middle")
torch.manual_seed(0)
model, _, _ = build_model(cfg)
model = model.cuda().train()
lb = model.lidar_backbone
datas = [build_inputs(cfg, batch, "kitti", 0, torch.device("cuda"), cycle=c)[1] for c in range(2)]
acc = collections.OrderedDict()
gpu = collections.OrderedDict()
REPS = 6
for it in range(REPS + 2):
    d = datas[it % 2]
    lidar = SparseTensor(d["lidar"].F, d["lidar"].C)
    n = lidar.F.shape[0]
    feats = torch.randn(n, 96, device="cuda")
    torch.cuda.synchronize()
    gen = lb.forward_steps(lidar, feats)
    t = time.perf_counter()
    ev = [torch.cuda.Event(enable_timing=True)]
    ev[0].record()
    names = []
    while True:
        try:
            tok = next(gen)
        except StopIteration as done:
            preds = done.value
            tok = "heads"
        now = time.perf_counter()
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(e)
        names.append(tok)
        if it >= 2:
            acc[len(names), tok] = acc.get((len(names), tok), 0.0) + (now - t) * 1e3
        t = now
        if tok == "heads":
            break
    loss = preds["lidar_seg_logit"].square().mean() + preds["lidar_seg_logit2"].square().mean()
    tb = time.perf_counter()
    eb0 = torch.cuda.Event(enable_timing=True); eb0.record()
    loss.backward()
    tb1 = time.perf_counter()
    eb1 = torch.cuda.Event(enable_timing=True); eb1.record()
    torch.cuda.synchronize()
    if it >= 2:
        for i, tok in enumerate(names):
            gpu[i + 1, tok] = gpu.get((i + 1, tok), 0.0) + ev[i].elapsed_time(ev[i + 1])
        acc["bwd", "backward"] = acc.get(("bwd", "backward"), 0.0) + (tb1 - tb) * 1e3
        gpu["bwd", "backward"] = gpu.get(("bwd", "backward"), 0.0) + eb0.elapsed_time(eb1)
    model.zero_grad(set_to_none=True)
print("LiDAR branch alone, batch %d: host issue ms | GPU span ms (events on the stream; a stage that is host-bound shows GPU span ~ host)" % batch)
th = tg = 0
for k in acc:
    print("  %-14s %7.2f | %7.2f" % (k[1], acc[k] / REPS, gpu[k] / REPS))
    if k[0] != "bwd":
        th += acc[k] / REPS; tg += gpu[k] / REPS
print("  %-14s %7.2f | %7.2f" % ("forward sum", th, tg))
