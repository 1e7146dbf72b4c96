import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fusiontransformer_amd import gemm_tuning
import torch.cuda.tunable as tunable
f = gemm_tuning.enable(0)
print("private file", f, "exists", os.path.exists(f), "lines", len(open(f).read().splitlines()) if os.path.exists(f) else 0)
print("results before any gemm:", len(tunable.get_results()))
a = torch.randn(2312, 768, device="cuda"); w = torch.randn(768, 768, device="cuda"); b = torch.randn(768, device="cuda")
torch.cuda.synchronize(); t = time.perf_counter()
y = torch.addmm(b, a, w.t()); torch.cuda.synchronize()
print("first addmm %.1f ms" % ((time.perf_counter() - t) * 1e3), "results now:", len(tunable.get_results()))
print([r for r in tunable.get_results()][:3])
print("file lines now", len(open(f).read().splitlines()))
print(open(f).read()[:600])
