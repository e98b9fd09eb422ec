# wall-clock A/B of an option on the bench workload (32 pairs per step, 1080p): python scripts/exp23.py name v0 v1
import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
name, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
import os
W, H, NP = 1920, 1080, int(os.environ.get("NP", "32"))
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, NP + 1, device=torch.device("cuda"))
flows = torch.empty((NP, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
for r in range(4):
    for v in vals:
        ctx.set_option(name, v)
        for _ in range(5): ctx.farneback_clip(frames, flows, **P)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40): ctx.farneback_clip(frames, flows, **P)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s=%d: %.1f us per pair (%.0f pairs/s)" % (name, v, dt / 40 / NP * 1e6, 40 * NP / dt), flush=True)
