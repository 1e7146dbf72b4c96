"""How fast is the sparse-conv tile kernel as a plain dense fp32 GEMM at the ViT linear shapes (ftx_rows_gemm, no gathers), next to the
library GEMM torch picks (TunableOp selections loaded)?  usage: python tools/probes/rows_gemm_vit_shapes.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusiontransformer_amd import functional as spf, gemm_tuning
gemm_tuning.enable(0, tune_missing=True)
L = spf._lib.load()

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for variant in (0, 1, 2):
    L.ftx_spconv_set_gemm_variant(variant)
    print("variant", variant)
    for M, K, N in [(2312, 768, 2304), (2312, 768, 768), (2312, 768, 3072), (2312, 3072, 768)]:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        t_own = timeit(lambda: L.ftx_rows_gemm(x.data_ptr(), M, w.data_ptr(), 1, b.data_ptr(), K, N, out.data_ptr(), spf.stream()))
        t_lib = timeit(lambda: torch.addmm(b, x, w.t()))
        ref = torch.addmm(b, x, w.t())
        err = (out - ref).abs().max().item()
        fl = 2.0 * M * K * N
        print("  %5d x %5d x %5d   own %7.1f us %6.1f TF   library %7.1f us %6.1f TF   max diff %.2e" % (M, K, N, t_own, fl / t_own / 1e6, t_lib, fl / t_lib / 1e6, err))
L.ftx_spconv_set_gemm_variant(0)
