import torch, time
def t(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
for mb in (128, 512, 2048):
    n = (mb << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    print("%5d MB  read-only (sum) %.0f GB/s | write-only (fill) %.0f GB/s | copy %.0f GB/s | a+b->c %.0f GB/s" % (
        mb, mb * 1.048576e6 / t(lambda: a.sum()) / 1e9, mb * 1.048576e6 / t(lambda: b.fill_(1.0)) / 1e9,
        2 * mb * 1.048576e6 / t(lambda: b.copy_(a)) / 1e9, 3 * mb * 1.048576e6 / t(lambda: torch.add(a, b, out=b)) / 1e9))
