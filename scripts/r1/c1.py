# BASELINE config 1 on the GPU: 640x480 translating-texture clip, reference parameters
import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 640, 480
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.translating_clip(W, H, 33, device=torch.device("cuda"))
flows = torch.empty((32, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
for _ in range(10): ctx.farneback_clip(frames, flows, **P)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): ctx.farneback_clip(frames, flows, **P)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50 / 32
core = flows[:, 40:-40, 40:-40]
print("C1 640x480: %.2f us/frame  %.0f fps; interior mean flow (%.4f, %.4f)" % (dt * 1e6, 1 / dt, float(core[..., 0].mean()), float(core[..., 1].mean())))
