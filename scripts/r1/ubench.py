import torch, time
def bw(nbytes, iters=20):
    a = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): b.copy_(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / iters
    return 2 * nbytes / dt / 1e9
for mb in (32, 64, 128, 256, 512, 1024, 2048):
    print("copy %5d MB: %.0f GB/s (read+write)" % (mb, bw(mb << 20)))
