# wall-clock A/B of builds of librcflow on the bench workload: python scripts/exp24.py libA libB ...  (each run in a subprocess via RCFLOW_LIB)
import os, subprocess, sys
code = r'''
import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H, NP = 1920, 1080, 32
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, NP + 1, device=torch.device("cuda"))
flows = torch.empty((NP, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
for _ in range(5): ctx.farneback_clip(frames, flows, **P)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): ctx.farneback_clip(frames, flows, **P)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%.2f us per pair  checksum %r" % (dt / 40 / NP * 1e6, float(flows.double().sum())))
'''
for r in range(3):
    for lib in sys.argv[1:]:
        env = dict(os.environ, RCFLOW_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(lib, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
