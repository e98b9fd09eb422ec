"""Same-process interleaved A/B of `ablate` option values on the bench workload (32 pairs of 1080p, Farnebäck only).
usage: abl_ab.py 0 1024 16384 ...   -> us per pair for each value, three rounds."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H, NP = 1920, 1080, 32
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
vals = [int(v) for v in sys.argv[1:]] or [0]
frames = synth.surf_clip(W, H, NP + 1, device=torch.device("cuda"))
flows = torch.empty((NP, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
ref = None
for r in range(3):
    for v in vals:
        ctx.set_option("ablate", v)
        for _ in range(5): ctx.farneback_clip(frames, flows, **P)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): ctx.farneback_clip(frames, flows, **P)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        cs = float(flows.double().sum())
        print("ablate %6d: %.2f us per pair  checksum %r" % (v, dt / 30 / NP * 1e6, cs), flush=True)
