"""Skinny dense GEMMs of the point branch (ftx_rows_gemm: nn.Linear on 81 k point rows, heads, 1x1x1 convs) and their weight gradients
(ftx_spconv_pairs_wgrad, dense mode) at the bench workload's shapes: microseconds against the HBM / MFMA roof of each.
usage (GPU box): python tools/bench_rows_gemm.py [--rows 81237]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusiontransformer_amd import functional as spf

ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=81237); ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
L = spf._lib.load()
n = args.rows
SHAPES = [(32, 256), (256, 128), (128, 96), (96, 256), (96, 20), (96, 32), (32, 64), (64, 128), (128, 256), (384, 256), (192, 128)]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


print("%4s -> %4s | %8s %8s | %8s %8s | %8s %8s | roof us (max of 4(ca+co)n / 8 TB/s, 2 n ca co / 157.3 TF)" % ("ca", "co", "fwd us", "x roof", "dgrad us", "x roof", "wgrad us", "x roof"))
for ca, co in SHAPES:
    rows = n if max(ca, co) <= 256 or ca == 96 else n // 8
    x = torch.randn(rows, ca, device="cuda"); w = torch.randn(co, ca, device="cuda") * 0.05; b = torch.zeros(co, device="cuda")
    y = torch.empty(rows, co, device="cuda"); gy = torch.randn(rows, co, device="cuda"); gx = torch.empty(rows, ca, device="cuda")
    roof = 1e6 * max(4.0 * rows * (ca + co) / 8e12, 2.0 * rows * ca * co / 157.3e12)
    t_f = timeit(lambda: L.ftx_rows_gemm(x.data_ptr(), rows, w.data_ptr(), 1, b.data_ptr(), ca, co, y.data_ptr(), spf.stream()))
    t_d = timeit(lambda: L.ftx_rows_gemm(gy.data_ptr(), rows, w.data_ptr(), 0, 0, co, ca, gx.data_ptr(), spf.stream()))
    ws_bytes = int(L.ftx_spconv_pairs_wgrad_workspace_bytes(rows, co, ca, 1)); ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    dw = torch.empty(co, ca, device="cuda")
    t_w = timeit(lambda: L.ftx_spconv_pairs_wgrad(gy.data_ptr(), rows, 0, x.data_ptr(), rows, 0, 0, rows, co, ca, 1, dw.data_ptr(), ws.data_ptr(), ws_bytes, spf.stream()))
    print("%4d -> %4d | %8.1f %8.2f | %8.1f %8.2f | %8.1f %8.2f | %6.1f  (%d rows)" % (ca, co, t_f, t_f / roof, t_d, t_d / roof, t_w, t_w / roof, roof, rows))
