# frame-at-a-time wall clock (push_frame, one 1080p frame per call) for builds given as arguments (RCFLOW_LIB)
import os, subprocess, sys
code = r'''
import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
dev = torch.device("cuda")
frames = synth.surf_clip(W, H, 8, device=dev)
ctx = Context(W, H)
flow = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
for i in range(50): ctx.push_frame(frames[i % 8], flow, **P)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 400
for i in range(n): ctx.push_frame(frames[i % 8], flow, **P)
torch.cuda.synchronize()
print("%.1f us per frame" % ((time.perf_counter() - t0) / n * 1e6))
'''
for r in range(3):
    for lib in sys.argv[1:]:
        env = dict(os.environ, RCFLOW_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(lib, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
