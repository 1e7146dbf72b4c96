"""Micro-benchmark of the fused attention kernels at the ViT's shape (578 tokens, 12 heads) over the built tilings.
usage (GPU box): python tools/bench_attn.py [--batches 1,2,4,8] [--iters 50]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusiontransformer_amd import functional as spf

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="1,2,4,8")
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--tokens", type=int, default=578)
ap.add_argument("--heads", type=int, default=12)
args = ap.parse_args()
L = spf._lib.load()
T, H = args.tokens, args.heads
CFGS = [(0, 0), (4, 2), (2, 2), (2, 4), (1, 2), (1, 4), (1, 8)]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


print("%5s %8s | %9s %9s | %9s %9s | max |d| vs (4,2): out, grad" % ("batch", "(qw,spl)", "fwd us", "TFLOP/s", "bwd us", "TFLOP/s"))
for B in [int(x) for x in args.batches.split(",")]:
    g = torch.Generator(device="cuda"); g.manual_seed(B)
    qkv = torch.randn(B, T, 3, H, 64, device="cuda", generator=g)
    go = torch.randn(B, T, H * 64, device="cuda", generator=g)
    out = torch.empty(B, T, H * 64, device="cuda"); lse = torch.empty(B, H, T, device="cuda")
    gq = torch.empty_like(qkv)
    ws_bytes = int(L.ftx_attn_bwd_workspace_bytes(B, T, H)); ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    flop_f = 4.0 * B * H * T * T * 64
    ref = None
    for qw, sp in CFGS:
        f = lambda: L.ftx_attn_fwd_tiled(qkv.data_ptr(), B, T, H, 64, 0.125, out.data_ptr(), lse.data_ptr(), qw, sp, spf.stream())
        bw = lambda: L.ftx_attn_bwd_tiled(qkv.data_ptr(), out.data_ptr(), go.data_ptr(), lse.data_ptr(), B, T, H, 64, 0.125, gq.data_ptr(), ws.data_ptr(), ws_bytes, qw, sp, spf.stream())
        tf = timeit(f); tb = timeit(bw)
        if (qw, sp) == (4, 2):
            ref = (out.clone(), gq.clone())
        d = "" if ref is None else "%.2e %.2e" % ((out - ref[0]).abs().max().item(), (gq - ref[1]).abs().max().item())
        print("%5d %8s | %9.1f %9.1f | %9.1f %9.1f | %s" % (B, "auto" if qw == 0 else "(%d,%d)" % (qw, sp), tf, flop_f / tf / 1e6, tb, 2.5 * flop_f / tb / 1e6, d))
