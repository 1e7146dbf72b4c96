"""Diagnostic: product (fp32, GPU) and oracle (fp32, CPU) gradients both measured against the
oracle run in float64.  Separates rounding chaos (ReLU gates, train-mode BN) from real bugs."""
import json, os, sys, copy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import oracle_inputs, product_inputs, small_cfg
from fusiontransformer_amd.data.synth import make_batch
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.trainer import fusion_losses
from oracle import ft_oracle as O

cfg = small_cfg("middle")
torch.manual_seed(1)
oracle = O.build_model(dict(cfg.MODEL))
model, _, _ = build_model(cfg)
model.load_state_dict(oracle.state_dict()); model = model.cuda()
oracle64 = copy.deepcopy(oracle).double()
batch = make_batch([2, 3], max_points=2000)
with torch.no_grad():
    oracle.eval(); oracle(oracle_inputs(batch)); oracle.train()
li = oracle.lidar_backbone.last_index
g = torch.Generator().manual_seed(5)
masks = {"y1": (torch.rand(li["x4"].C.shape[0], 256, generator=g) > 0.3).float(), "y3": (torch.rand(li["x2"].C.shape[0], 128, generator=g) > 0.3).float()}
oracle.lidar_backbone.dropout_masks = masks
oracle64.lidar_backbone.dropout_masks = {k: v.double() for k, v in masks.items()}
model.lidar_backbone.dropout_masks = {k: v.cuda() for k, v in masks.items()}
cw = torch.tensor(cfg.TRAIN.CLASS_WEIGHTS)
lab = torch.from_numpy(batch["seg_label"])
oracle.train(); oracle64.train(); model.train()
ref = oracle(oracle_inputs(batch)); l2, l3 = O.fusion_losses(ref, lab, cw, 0.1, True); (l2 + l3).backward()
i64 = oracle_inputs(batch); i64["img"] = i64["img"].double(); i64["lidar"].F = i64["lidar"].F.double()
r64 = oracle64(i64); a, b = O.fusion_losses(r64, lab, cw.double(), 0.1, True); (a + b).backward()
pin = product_inputs(batch); out = model(pin); p2, p3 = fusion_losses(out, pin["seg_label"], cw.cuda(), 0.1, True); (p2 + p3).backward()
for k in ref:
    print(k, "oracle32-vs-64 %.3e" % (ref[k].double() - r64[k]).abs().max().item(), " product-vs-64 %.3e" % (out[k].detach().cpu().double() - r64[k]).abs().max().item(),
          " product-vs-oracle32 %.3e" % (out[k].detach().cpu() - ref[k]).abs().max().item())
p32, p64, pp = dict(oracle.named_parameters()), dict(oracle64.named_parameters()), dict(model.named_parameters())
rows = []
for n, p in p64.items():
    if p.grad is None: continue
    nr = p.grad.norm().item()
    e_o = (p32[n].grad.double() - p.grad).norm().item() / (nr + 1e-30)
    e_p = (pp[n].grad.cpu().double() - p.grad).norm().item() / (nr + 1e-30)
    rows.append((e_p, e_o, nr, n))
rows.sort(reverse=True)
print("worst product-vs-fp64 L2-rel | oracle32-vs-fp64 L2-rel | ||g64|| | name")
for r in rows[:25]: print("%.3e  %.3e  %.3e  %s" % r)
print("median product %.3e  median oracle32 %.3e" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
