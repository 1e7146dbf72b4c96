"""Per-shape micro-benchmark of the BatchNorm kernels on the (rows, channels) of the bench workload (batch 4): microseconds and GB/s of the
forward (statistics + apply) and of the backward (statistics + apply), against the bytes each must move.
usage (GPU box): python tools/bench_bn.py [--iters 30]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusiontransformer_amd import functional as spf

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
L = spf._lib.load()
# (rows, channels, how many such BatchNorms in one training step of the middle-fusion model at batch 4, relu, residual)
SHAPES = [(81237, 32, 2, 1, 0), (81237, 96, 5, 1, 0), (81237, 96, 2, 1, 1), (81237, 256, 2, 1, 0), (81237, 128, 1, 1, 0),
          (43016, 32, 3, 1, 0), (43016, 32, 2, 1, 1), (43016, 96, 3, 1, 0), (43016, 96, 2, 1, 1), (20197, 64, 4, 1, 0), (20197, 128, 5, 1, 0),
          (8102, 128, 4, 1, 0), (8102, 256, 5, 1, 0), (2949, 256, 8, 1, 0)]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


print("%8s %5s %3s | %8s %8s | %8s %8s" % ("rows", "c", "n", "fwd us", "GB/s", "bwd us", "GB/s"))
tf = tb = bf = bb = 0.0
for n, c, count, relu, res in SHAPES:
    x = torch.randn(n, c, device="cuda"); r = torch.randn(n, c, device="cuda") if res else None
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    y = torch.empty_like(x); mean = torch.empty(c, device="cuda"); inv = torch.empty(c, device="cuda")
    ws_bytes = int(L.ftx_bn_workspace_bytes(n, c)); ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    st = spf._stream_scratch()
    fwd = lambda: L.ftx_bn_train_fwd(x.data_ptr(), spf.ptr(r), g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, n, c, relu, y.data_ptr(),
                                     mean.data_ptr(), inv.data_ptr(), ws.data_ptr(), ws_bytes, st)
    gy = torch.randn(n, c, device="cuda"); gx = torch.empty_like(x); gres = torch.empty_like(x) if res else None
    gg, gb = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    bwd = lambda: L.ftx_bn_train_bwd(gy.data_ptr(), x.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), mean.data_ptr(), inv.data_ptr(), n, c, relu, gx.data_ptr(), spf.ptr(gres),
                                     gg.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws_bytes, st)
    t_f, t_b = timeit(fwd), timeit(bwd)
    by_f = 4.0 * n * c * (2 + (1 if res else 0) + 1)             # x twice (+ residual), y out
    by_b = 4.0 * n * c * (2 * (2 + (1 if (relu and res) else 0)) + 1 + (1 if res else 0))  # two passes over gy, x (, y when a residual went into it), gx (+ gres) out
    print("%8d %5d %3d | %8.1f %8.0f | %8.1f %8.0f" % (n, c, count, t_f, by_f / t_f / 1e3, t_b, by_b / t_b / 1e3))
    tf += count * t_f; tb += count * t_b; bf += count * by_f; bb += count * by_b
print("per step (weighted by count): fwd %.0f us at %.0f GB/s, bwd %.0f us at %.0f GB/s" % (tf, bf / tf / 1e3, tb, bb / tb / 1e3))
