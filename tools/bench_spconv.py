"""Per-layer micro-benchmark of the sparse-conv kernels on the bench workload's real kernel maps.
usage (GPU box): python tools/bench_spconv.py [--batch 4]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusiontransformer_amd import functional as spf
from fusiontransformer_amd.data.synth import make_batch
from fusiontransformer_amd.models.utils import initial_voxelize
from fusiontransformer_amd.sparse import PointTensor

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=4); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--filter", default="", help="only layers whose name contains one of these comma-separated substrings")
ap.add_argument("--what", default="gemm,reduce,wgrad,ostat", help="kernels to run (profiling runs: --what wgrad --iters 1)")
args = ap.parse_args()
WHAT = set(args.what.split(","))
b = make_batch(list(range(args.batch)))
z = PointTensor(torch.from_numpy(b["feats"]).cuda(), torch.from_numpy(b["coords"]).float().cuda())
x0 = initial_voxelize(z, 1, 1)
cm = x0.cm
layers = []  # (name, ks, cur_stride, stride, ca, co)
for lvl, (s, chans) in enumerate([(1, [(32, 32), (128, 96), (96, 96)]), (2, [(32, 32), (128, 96), (96, 96)]), (4, [(32, 64), (64, 64), (192, 128), (128, 128)]),
                                   (8, [(64, 128), (128, 128), (384, 256), (256, 256)]), (16, [(128, 256), (256, 256)])]):
    if s > 1:
        cm.kernel_map(2, s // 2, 2)
    for ca, co in chans:
        layers.append(("k3 s%d %d->%d" % (s, ca, co), 3, s, 1, ca, co))
for s, c in [(1, 32), (2, 32), (4, 64), (8, 128)]:
    layers.append(("k2 down s%d %d->%d" % (s, c, c), 2, s, 2, c, c))
layers.insert(0, ("k3 s1 4->32 (stem)", 3, 1, 1, 4, 32))
HBM, MFMA = 8.0e12, 157.3e12     # peaks the roofs are priced against (MI355X_MICROARCH.md)

def timeit(fn, on=True):
    if not on:
        return float("nan")
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3

print("%-24s %9s %9s | %8s %8s %8s %8s %8s | %7s %7s | %8s %9s" % ("layer", "rows", "pairs", "gemm us", "reduce", "ostat", "ostat dg", "wgrad", "gemmTF", "wgradTF", "roof us", "fwd/roof"))
tot = [0, 0, 0]
tot_os = [0.0, 0.0, 0.0, 0.0]    # forward as shipped (ostat where supported), the same layers on the pair path, roof of forward, roof of wgrad
for name, ks, cur, st, ca, co in layers:
    if args.filter and not any(f in name for f in args.filter.split(",")):
        continue
    km = cm.kernel_map(ks, cur, st)
    A = torch.randn(km.n_in, ca, device="cuda"); W = torch.randn(ks ** 3, ca, co, device="cuda") * 0.05
    G = torch.randn(km.n_out, co, device="cuda")
    L = spf._lib.load()
    tmp = torch.empty(km.n_pairs, co, device="cuda"); out = torch.empty(km.n_out, co, device="cuda")
    t_g = timeit(lambda: L.ftx_spconv_pairs_gemm(A.data_ptr(), km.n_in, km.pair_in.data_ptr(), W.data_ptr(), int(os.environ.get("FTX_BENCH_WT", "0")), km.koff.data_ptr(), km.n_pairs, ca, co, ks ** 3, tmp.data_ptr(), spf.stream()), "gemm" in WHAT)
    t_r = timeit(lambda: L.ftx_spconv_reduce(tmp.data_ptr(), km.pos.data_ptr(), km.n_out, co, ks ** 3, out.data_ptr(), spf.stream()), "reduce" in WHAT)
    ws_bytes = int(L.ftx_spconv_pairs_wgrad_workspace_bytes(km.n_pairs, ca, co, ks ** 3))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda"); dW = torch.empty(ks ** 3, ca, co, device="cuda")
    t_w = timeit(lambda: L.ftx_spconv_pairs_wgrad(A.data_ptr(), km.n_in, km.pair_in.data_ptr(), G.data_ptr(), km.n_out, km.pair_out.data_ptr(), km.koff.data_ptr(),
                                                  km.n_pairs, ca, co, ks ** 3, dW.data_ptr(), ws.data_ptr(), ws_bytes, spf.stream()), "wgrad" in WHAT)
    fl = 2.0 * km.n_pairs * ca * co
    t_o = t_od = float("nan")
    if "ostat" in WHAT and spf.ostat_supported(ca, co, ks ** 3):
        t_o = timeit(lambda: L.ftx_spconv_ostat(A.data_ptr(), km.n_in, km.nbr.data_ptr(), km.n_out, W.data_ptr(), 0, 0, ca, co, ks ** 3, out.data_ptr(), 0, 0, spf.stream()))
    if "ostat" in WHAT and km.submanifold and spf.ostat_supported(co, ca, ks ** 3, True):
        gin = torch.empty(km.n_in, ca, device="cuda")
        t_od = timeit(lambda: L.ftx_spconv_ostat(G.data_ptr(), km.n_out, km.nbr.data_ptr(), km.n_in, W.data_ptr(), 1, 1, co, ca, ks ** 3, gin.data_ptr(), 0, 0, spf.stream()))
    roof = 1e6 * max((4.0 * km.n_pairs * (ca + co) + 4.0 * ks ** 3 * ca * co) / HBM, fl / MFMA)     # the launch's binding roof, us
    fwd = t_o if (t_o == t_o and spf.ostat_preferred(ca, co, ks ** 3)) else t_g + t_r     # what the host wrapper launches for this layer
    print("%-24s %9d %9d | %8.1f %8.1f %8.1f %8.1f %8.1f | %7.1f %7.1f | %8.1f %9.2f" % (name, km.n_out, km.n_pairs, t_g, t_r, t_o, t_od, t_w, fl / t_g / 1e6, fl / t_w / 1e6, roof, roof / fwd))
    tot[0] += t_g; tot[1] += t_r; tot[2] += t_w
    tot_os[0] += fwd; tot_os[1] += t_g + t_r; tot_os[2] += roof; tot_os[3] += roof
print("sum: gemm %.0f us  reduce %.0f us  wgrad %.0f us" % tuple(tot))
print("forward as shipped (ostat where it applies) %.0f us, all on the pair-list path %.0f us" % (tot_os[0], tot_os[1]))
print("binding-roof fraction over these layers, forward + weight gradient: %.3f (sum of max(bytes / 8 TB/s, flop / 157.3 TF) / sum of time)"
      % ((tot_os[2] + tot_os[3]) / (tot_os[0] + tot[2])))
