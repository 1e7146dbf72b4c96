"""Host cost of one kernel launch through each layer of the stack, on an otherwise idle GPU (the queue never fills: a synchronise every
200 launches): the C-ABI entry through ctypes with ready-made arguments, the functional wrapper (validation + output allocation), and the
autograd Function around it.  usage: python tools/probes/launch_cost.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusiontransformer_amd import _lib, functional as spf

lib = _lib.load()
dev = torch.device("cuda")
n, c = 2048, 32
x = torch.randn(n, c, device=dev)
w = torch.randn(c, c, device=dev)
out = torch.empty(n, c, device=dev)
st = _lib.stream()

def timeit(fn, reps=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t = 0.0
    done = 0
    while done < reps:
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        t += time.perf_counter() - t0
        torch.cuda.synchronize()
        done += 200
    return t / done * 1e6

args = (x.data_ptr(), n, w.data_ptr(), 0, 0, c, c, out.data_ptr(), st)
print("ftx_rows_gemm via ctypes, arguments ready      %.2f us / launch" % timeit(lambda: lib.ftx_rows_gemm(*args)))
print("  + data_ptr() x3 and the stream lookup          %.2f" % timeit(lambda: lib.ftx_rows_gemm(x.data_ptr(), n, w.data_ptr(), 0, 0, c, c, out.data_ptr(), _lib.stream())))
print("torch.empty((n, c), device)                     %.2f" % timeit(lambda: torch.empty((n, c), device=dev)))
print("spf rows-gemm wrapper (validation + allocation)  %.2f" % timeit(lambda: spf._rows_gemm(x, w, False, None, c)))
xr = x.clone().requires_grad_(True)
wr = w.clone().requires_grad_(True)
class Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return spf._rows_gemm(a, b, False, None, c)
    @staticmethod
    def backward(ctx, g):
        return None, None
print("the same inside a torch.autograd.Function.apply    %.2f" % timeit(lambda: Fn.apply(xr, wr)))
print("torch.nn.functional.linear (library GEMM), no grad %.2f" % timeit(lambda: torch.nn.functional.linear(x, w)))
print("torch add (elementwise), no grad                 %.2f" % timeit(lambda: torch.add(x, x)))
print("torch add, grad recorded                         %.2f" % timeit(lambda: torch.add(xr, xr)))
