# one-off soak: the register-resident flow kernel vs the LDS-resident form vs one launch per
# iteration on random clips of random sizes; any race or uninitialised read shows up as a bit difference
import sys, torch, numpy as np
sys.path.insert(0, '.')
from ripcurrents_amd.api import Context
ctx = Context(2048, 1200)
rng = np.random.RandomState(0)
bad = 0
for trial in range(60):
    w, h = int(rng.randint(33, 700)), int(rng.randint(33, 500))
    if trial % 2: w, h = w & ~3, h & ~3          # exact half / quarter sizes: the fused pyramid path
    T = int(rng.randint(2, 7))
    base = rng.randint(0, 256, size=(h + 16, w + 16)).astype(np.float32)
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(base, 1.5 + 2 * rng.rand()) * 3 % 256
    clip = np.stack([base[8 + (t * 3) % 5:8 + (t * 3) % 5 + h, 8 + t:8 + t + w] for t in range(T)]).astype(np.uint8)
    d = torch.as_tensor(clip).cuda()
    flags = 256 if trial % 3 == 0 else 0
    P = dict(pyr_scale=0.5, levels=int(rng.randint(0, 4)), winsize=3, iterations=2 if trial % 4 else 4, poly_n=15, poly_sigma=1.2, flags=flags)
    outs = []
    for opts in ({"ablate": 0, "fuse_iters": 1}, {"ablate": 64, "fuse_iters": 1}, {"ablate": 0, "fuse_iters": 0}, {"ablate": 0, "fuse_iters": 1, "fuse_pyr": 0}, {"ablate": 0, "fuse_iters": 1, "fuse_pyr": 1}):
        for k, v in opts.items(): ctx.set_option(k, v)
        out = torch.empty((T - 1, h, w, 2), dtype=torch.float32, device="cuda")
        ctx.farneback_clip(d, out, **P)
        outs.append(out.cpu().numpy())
    same = all(np.array_equal(outs[0], o) for o in outs[1:])
    if not same:
        bad += 1
        print("MISMATCH trial", trial, w, h, T, P, [float(np.abs(outs[0] - o).max()) for o in outs[1:]], flush=True)
ctx.set_option("ablate", 0); ctx.set_option("fuse_iters", 1)
print("trials 60, mismatches", bad)
# the strip-sweep kernel of the Gaussian winsize 10 / 20 call sites against the tile kernel
bad2 = 0
for trial in range(24):
    w, h = int(rng.randint(9, 500)), int(rng.randint(9, 400))
    T = int(rng.randint(2, 5))
    clip = rng.randint(0, 256, size=(T, h, w)).astype(np.float32)
    clip = np.stack([gaussian_filter(c, 2.0) for c in clip]).astype(np.uint8)
    d = torch.as_tensor(clip).cuda()
    P = dict(pyr_scale=0.5, levels=int(rng.randint(0, 3)), winsize=10 if trial % 2 else 20, iterations=int(rng.randint(1, 4)), poly_n=15, poly_sigma=1.2, flags=256)
    outs = []
    for abl in (65536, 8388608, 8388608 + 131072):
        ctx.set_option("ablate", abl)
        out = torch.empty((T - 1, h, w, 2), dtype=torch.float32, device="cuda")
        ctx.farneback_clip(d, out, **P)
        outs.append(out.cpu().numpy())
    if not all(np.array_equal(outs[0], o) for o in outs[1:]):
        bad2 += 1
        print("SWEEP MISMATCH trial", trial, w, h, T, P, flush=True)
ctx.set_option("ablate", 0)
print("sweep trials 24, mismatches", bad2)
