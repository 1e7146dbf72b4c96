import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from fusiontransformer_amd.data.synth import make_batch
from fusiontransformer_amd.trainer import fusion_losses
from helpers import product_inputs
from test_model_gpu import _pair
modes = [m == "1" for m in sys.argv[1]]
cfg, oracle, model, _ = _pair("middle", seed=4)
model.train()
pin = product_inputs(make_batch([6, 7], max_points=3000))
res = []
for overlap in modes:
    model.overlap_branches = overlap
    model.zero_grad(set_to_none=True)
    torch.manual_seed(0)
    out = model(pin)
    l2, l3 = fusion_losses(out, pin["seg_label"], None, 0.1, True)
    (l2 + l3).backward()
    torch.cuda.synchronize()
    res.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
for i in range(1, len(res)):
    bad = [(n, (res[i][n] - res[0][n]).abs().max().item()) for n in res[i] if not torch.equal(res[i][n], res[0][n])]
    print("run", i, "overlap", modes[i], "differing params:", len(bad), bad[:6])
