import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusiontransformer_amd import functional as spf
from fusiontransformer_amd.data.synth import make_batch
from fusiontransformer_amd.models.utils import initial_voxelize
from fusiontransformer_amd.sparse import PointTensor
b = make_batch(list(range(4)))
z = PointTensor(torch.from_numpy(b["feats"]).cuda(), torch.from_numpy(b["coords"]).float().cuda())
x0 = initial_voxelize(z, 1, 1); cm = x0.cm
km = cm.kernel_map(3, 1, 1)
ca, co = 128, 96
A = torch.randn(km.n_in, ca, device="cuda"); W = torch.randn(27, ca, co, device="cuda") * 0.05
tmp = torch.empty(km.n_pairs, co, device="cuda")
L = spf._lib.load()
raw = ctypes.CDLL(os.path.join(os.path.dirname(spf.__file__), "csrc", "libftx.so"))
for it in range(3):
    L.ftx_spconv_pairs_gemm(A.data_ptr(), km.n_in, km.pair_in.data_ptr(), W.data_ptr(), 0, km.koff.data_ptr(), km.n_pairs, ca, co, 27, tmp.data_ptr(), spf.stream())
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (32 * 64))()
raw.ftx_debug_ts(buf)
ts = np.array(buf[:], dtype=np.int64).reshape(32, 64)
ok = ts[:, 0] > 0
d = np.diff(ts[ok][:, :60], axis=1)
print("blocks", ok.sum())
names = ["wait+bar", "stores", "dma", "mfma", "next"]
for st in range(12):
    print("step %2d: " % (st + 8) + "  ".join("%s %6.0f" % (names[j], np.median(d[:, st * 5 + j])) for j in range(5) if st * 5 + j < d.shape[1]))
