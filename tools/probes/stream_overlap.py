"""Do a library GEMM of the ViT and a libftx kernel overlap when issued on two streams?  Each pair: time A alone, B alone, and A on one stream
with B on another (queued 60 deep each).  Perfect overlap = max(A, B); none = A + B."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusiontransformer_amd import functional as spf
from fusiontransformer_amd.data.synth import make_batch
from fusiontransformer_amd.models.utils import initial_voxelize
from fusiontransformer_amd.sparse import PointTensor

N = 60
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(2312, 768, device="cuda"); w = torch.randn(3072, 768, device="cuda"); y = torch.empty(2312, 3072, device="cuda")
gemm = lambda: torch.mm(x, w.t(), out=y)

b = make_batch(list(range(4)))
z = PointTensor(torch.from_numpy(b["feats"]).cuda(), torch.from_numpy(b["coords"]).float().cuda())
cm = initial_voxelize(z, 1, 1).cm
km = cm.kernel_map(3, 1, 1)
L = spf._lib.load()
A = torch.randn(km.n_in, 128, device="cuda"); W = torch.randn(27, 128, 96, device="cuda") * 0.05
tmp = torch.empty(km.n_pairs, 96, device="cuda"); out = torch.empty(km.n_out, 96, device="cuda")
def pg(): L.ftx_spconv_pairs_gemm(A.data_ptr(), km.n_in, km.pair_in.data_ptr(), W.data_ptr(), 0, km.koff.data_ptr(), km.n_pairs, 128, 96, 27, tmp.data_ptr(), spf.stream())
def red(): L.ftx_spconv_reduce(tmp.data_ptr(), km.pos.data_ptr(), km.n_out, 96, 27, out.data_ptr(), spf.stream())
xb = torch.randn(km.n_out, 96, device="cuda"); g = torch.ones(96, device="cuda"); be = torch.zeros(96, device="cuda")
def bn():
    with torch.no_grad():
        spf.batch_norm(xb, g, be, None, None, True, 0.1, 1e-5, relu=True)
qkv = torch.randn(4, 578, 3, 12, 64, device="cuda"); ao = torch.empty(4, 578, 768, device="cuda"); lse = torch.empty(4, 12, 578, device="cuda")
def attn(): L.ftx_attn_fwd(qkv.data_ptr(), 4, 578, 12, 64, 0.125, ao.data_ptr(), lse.data_ptr(), spf.stream())

def run(fa, fb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa is not None:
        with torch.cuda.stream(s1):
            for _ in range(N): fa()
    if fb is not None:
        with torch.cuda.stream(s2):
            for _ in range(N): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6

def interleaved(fa, fb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        with torch.cuda.stream(s1): fa()
        with torch.cuda.stream(s2): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6

for _ in range(3):
    gemm(); pg(); red(); bn(); attn()
print("%-34s %8s %8s %10s %10s   (us per pair of launches)" % ("pair (A | B)", "A alone", "B alone", "2 streams", "interleaved"))
for name, fa, fb in [("library GEMM | pairs_gemm", gemm, pg), ("library GEMM | spconv_reduce", gemm, red), ("library GEMM | batch_norm (2 launches)", gemm, bn),
                     ("library GEMM | library GEMM", gemm, gemm), ("attn_fwd | pairs_gemm", attn, pg), ("pairs_gemm | spconv_reduce", pg, red)]:
    a, bt = run(fa, None), run(None, fb)
    print("%-34s %8.1f %8.1f %10.1f %10.1f" % (name, a, bt, run(fa, fb), interleaved(fa, fb)))
