import torch, time
torch.manual_seed(0)
dev = 'cuda'
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def split3(x):
    hi = x.to(torch.bfloat16); r = x - hi.float()
    mid = r.to(torch.bfloat16); r = r - mid.float()
    lo = r.to(torch.bfloat16)
    return hi, mid, lo
for (M, K, N) in [(2312, 768, 2304), (2312, 768, 768), (2312, 768, 3072), (2312, 3072, 768), (9248, 768, 2304)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev)
    ref = a.double() @ b.double()
    t32 = bench(lambda: a @ b)
    e32 = ((a @ b).double() - ref).abs().max().item() / ref.abs().max().item()
    ah, am, al = split3(a); bh, bm, bl = split3(b)
    A6 = torch.cat([ah, ah, ah, am, am, al], 1).contiguous()
    B6 = torch.cat([bh, bm, bl, bh, bm, bh], 0).contiguous()
    def f6():
        return torch.mm(A6, B6, out_dtype=torch.float32) if False else (A6 @ B6)
    t6 = bench(f6)
    # accuracy with fp32 accumulation: do the 6 products as separate bf16 matmuls accumulating in fp32 is not available via @ (bf16 out),
    # so measure time only here; accuracy via float emulation
    emu = sum((x.float().double() @ y.float().double()) for x, y in [(ah,bh),(ah,bm),(ah,bl),(am,bh),(am,bm),(al,bh)])
    e6 = (emu - ref).abs().max().item() / ref.abs().max().item()
    print(f"M{M} K{K} N{N}: fp32 {t32:.1f} us ({2*M*K*N/t32/1e6:.1f} TF)  bf16 K'=6K {t6:.1f} us ({2*M*6*K*N/t6/1e6:.1f} TF bf16, {2*M*K*N/t6/1e6:.1f} TF eff)  err32 {e32:.2e} split-exact-err {e6:.2e}", flush=True)
