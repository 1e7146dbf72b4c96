import os, sys, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusiontransformer_amd.models import transformers as T
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, w)
        ctx.shape = x.shape
        return torch.addmm(b, x2, w.t()).view(*x.shape[:-1], w.shape[0])
    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        ones = torch.ones(1, dy2.shape[0], device=dy.device, dtype=dy.dtype)
        return (dy2 @ w).view(ctx.shape), dy2.t() @ x2, (ones @ dy2).view(-1)
class Lin(nn.Linear):
    def forward(self, x):
        return _LinearFn.apply(x, self.weight, self.bias)
torch.manual_seed(0)
for use_custom in (False, True):
    m = nn.Sequential(T.Block(768, 12)).cuda()
    m[0].attn.attn_impl = "torch"
    if use_custom:
        for mod in [m[0].attn.qkv, m[0].attn.proj, m[0].mlp.fc1, m[0].mlp.fc2]:
            mod.__class__ = Lin
    x = torch.randn(2, 578, 768, device="cuda", requires_grad=True)
    g = torch.cuda.make_graphed_callables(m, (x.detach().clone().requires_grad_(True),))
    params = list(m.parameters()); names = [n for n, _ in m.named_parameters()]
    res = []
    for it in range(3):
        y = g(x); gr = torch.autograd.grad((y * y).mean(), params); torch.cuda.synchronize(); res.append([t.clone() for t in gr])
    print("custom linear" if use_custom else "nn.Linear", "replay1!=replay0:", [n for n, a, b in zip(names, res[0], res[1]) if not torch.equal(a, b)], "replay2!=replay1:", [n for n, a, b in zip(names, res[1], res[2]) if not torch.equal(a, b)])
