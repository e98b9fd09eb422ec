# single-frame histogram launch shape + thresholds timing at 1080p
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
dev = torch.device("cuda")
W, H = 1920, 1080
ctx = Context(W, H)
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
fr = synth.surf_clip(W, H, 2, device=dev)
f0 = ctx.calcOpticalFlowFarneback(fr[0], fr[1], None, **P)     # a real (textured) flow field
ctx.analysis_reset(W, H)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for hb in (0, 512, 1024, 2048, 4096, 8192, 0):
    ctx.set_option("hist_blocks", hb)
    print("hist_blocks=%5d  histogram %.1f us" % (hb, t(lambda: ctx.histogram_accumulate(f0))))
print("thresholds %.1f us" % t(lambda: ctx.thresholds()))
