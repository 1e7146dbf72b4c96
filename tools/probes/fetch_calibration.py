"""Known-byte calibration of the FETCH_SIZE counter for THIS library's gather pattern (MI355X_MICROARCH.md: only wide streaming reads are
calibrated).  ftx_spconv_reduce with one offset is a pure row gather: out[r] = tmp[pos[r]], every row of `tmp` read exactly once in a random
order, rows of 128 B (32 channels) and 512 B (128 channels), table of 2 GiB (>> the 256 MiB Infinity Cache).  Run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 tools/probes/fetch_calibration.py
and read the counter of the spconv_reduce dispatches with tools/pmc_sq.py DIR spconv_reduce."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusiontransformer_amd import functional as spf

L = spf._lib.load()
for co in (32, 128):
    n = (2 << 30) // (co * 4)
    tmp = torch.empty(n, co, device="cuda").normal_()
    out = torch.empty(n, co, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    pos = torch.randperm(n, device="cuda", generator=g).to(torch.int32).view(1, n).contiguous()
    seq = torch.arange(n, device="cuda", dtype=torch.int32).view(1, n).contiguous()
    for name, p in (("random", pos), ("sequential", seq)):
        assert L.ftx_spconv_reduce(tmp.data_ptr(), p.data_ptr(), n, co, 1, out.data_ptr(), spf.stream()) == 0
        torch.cuda.synchronize()
        print("co=%d rows of %d B, %s order: %d rows, table bytes %d (+ %d index bytes)" % (co, co * 4, name, n, n * co * 4, n * 4), flush=True)
    assert torch.equal(out, tmp)
    del tmp, out, pos, seq
