import os, sys, faulthandler, torch, torch.nn as nn
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusiontransformer_amd.models import transformers as T
torch.manual_seed(0)
m = nn.Sequential(T.Block(768, 12)).cuda()
m[0].attn.attn_impl = "torch"
which = sys.argv[1]
x1 = torch.randn(2, 578, 768, device="cuda", requires_grad=True)
if which == "eager_then_capture":
    y = m(x1); (y * y).mean().backward(); torch.cuda.synchronize(); print("eager backward done", flush=True)
    m.zero_grad(set_to_none=True)
seg = T._TrunkSegment.__new__(T._TrunkSegment); nn.Module.__init__(seg); seg.embed = False; seg.blocks = nn.ModuleList([m[0]])
g1 = torch.cuda.make_graphed_callables(seg, (x1.detach().clone().requires_grad_(True),)); print("capture 1 done", flush=True)
y = g1(x1); (y * y).mean().backward(); torch.cuda.synchronize(); print("graphed backward done", flush=True)
x2 = torch.randn(1, 578, 768, device="cuda", requires_grad=True)
seg2 = T._TrunkSegment.__new__(T._TrunkSegment); nn.Module.__init__(seg2); seg2.embed = False; seg2.blocks = nn.ModuleList([m[0]])
g2 = torch.cuda.make_graphed_callables(seg2, (x2.detach().clone().requires_grad_(True),)); print("capture 2 (other shape, after a backward) done", flush=True)
y = g2(x2); (y * y).mean().backward(); torch.cuda.synchronize(); print("second graphed backward done", flush=True)
