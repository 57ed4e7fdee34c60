import torch, time
N, C = 100000, 64
a = torch.randn(N, C, device="cuda"); b = torch.randn(N, C, device="cuda")
def t(fn, k=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/k*1e6
ref = (a.double().t() @ b.double())
print("a.t()@b", t(lambda: a.t() @ b))
for B in (32, 64, 128, 256, 512, 1024):
    n2 = (N // B) * B
    def f():
        r = torch.bmm(a[:n2].view(B, N // B, C).transpose(1, 2), b[:n2].view(B, N // B, C)).sum(0)
        if n2 < N: r = r + a[n2:].t() @ b[n2:]
        return r
    print("bmm", B, t(f), float((f().double() - ref).abs().max() / ref.abs().max()))
print("einsum", t(lambda: torch.einsum("nj,nk->jk", a, b)))
