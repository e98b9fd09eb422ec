# host-pointer drop-in call (rcflow_farneback_u8, numpy in / numpy out): calls per second at 1080p
import sys, time, numpy as np
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
clip = synth.surf_clip(W, H, 4)
ctx = Context(W, H)
hf = np.empty((H, W, 2), np.float32)
ref = None
for pin in (0, 1):
    for i in range(5): ctx.calcOpticalFlowFarneback(clip[i % 3], clip[i % 3 + 1], hf, **P)
    t0 = time.perf_counter()
    n = 60
    for i in range(n): ctx.calcOpticalFlowFarneback(clip[i % 3], clip[i % 3 + 1], hf, **P)
    dt = (time.perf_counter() - t0) / n
    ctx.calcOpticalFlowFarneback(clip[0], clip[1], hf, **P)
    if ref is None: ref = hf.copy()
    print("run %d: %.0f us per call  %.0f pairs/s  same: %s" % (pin, dt * 1e6, 1 / dt, np.array_equal(ref, hf)), flush=True)
