# Farneback on hostile images (uncorrelated noise, constant, checkerboard, saturated steps): finite?, parity with the oracle
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from ripcurrents_amd.api import Context
w, h = 320, 240
ctx = Context(1920, 1080)
rng = np.random.RandomState(1)
yy, xx = np.mgrid[0:h, 0:w]
imgs = {
    "noise": (rng.randint(0, 256, (h, w)).astype(np.uint8), rng.randint(0, 256, (h, w)).astype(np.uint8)),
    "constant": (np.full((h, w), 77, np.uint8), np.full((h, w), 77, np.uint8)),
    "constant step": (np.full((h, w), 0, np.uint8), np.full((h, w), 255, np.uint8)),
    "checker 1px": ((((xx + yy) & 1) * 255).astype(np.uint8), (((xx + yy + 1) & 1) * 255).astype(np.uint8)),
    "vertical bars": (((xx // 4 & 1) * 255).astype(np.uint8), (((xx + 2) // 4 & 1) * 255).astype(np.uint8)),
    "half black": (np.where(xx < w // 2, 0, 255).astype(np.uint8), np.where(xx < w // 2 + 3, 0, 255).astype(np.uint8)),
}
P = [dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0),
     dict(pyr_scale=0.5, levels=2, winsize=20, iterations=3, poly_n=15, poly_sigma=1.2, flags=256),
     dict(pyr_scale=0.5, levels=3, winsize=5, iterations=3, poly_n=15, poly_sigma=1.2, flags=0)]
for name, (a, b) in imgs.items():
    for p in P:
        ref = orc.farneback(a, b, p["pyr_scale"], p["levels"], p["winsize"], p["iterations"], p["poly_n"], p["poly_sigma"], p["flags"])
        got = ctx.calcOpticalFlowFarneback(a, b, None, **p)
        got = got.cpu().numpy() if hasattr(got, "cpu") else np.asarray(got)
        err = np.abs(got - ref).max(-1)
        print("%-14s win %2d: finite gpu %s oracle %s  |flow| max gpu %.3g oracle %.3g  frac<=1e-3 %.4f  p50 %.2g" % (
            name, p["winsize"], bool(np.isfinite(got).all()), bool(np.isfinite(ref).all()), float(np.abs(got).max()), float(np.abs(ref).max()),
            float((err <= 1e-3).mean()), float(np.median(err))), flush=True)
# big noise frames through the batched path: no fault, finite
big = torch.randint(0, 256, (5, 1080, 1920), dtype=torch.uint8, device="cuda")
for p in P:
    out = ctx.farneback_clip(big, **p)
    torch.cuda.synchronize()
    print("1080p noise clip win %d: finite %s max |flow| %.3g" % (p["winsize"], bool(torch.isfinite(out).all()), float(out.abs().max())), flush=True)
