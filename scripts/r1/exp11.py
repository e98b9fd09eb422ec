# per-kernel event times: 4K 5-scale clip, and the per-frame analysis at 1080p
import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
def show(ctx, n, tag):
    rows = ctx.profile_read(); ctx.profile_enable(False)
    tot = sum(r["total_ms"] for r in rows)
    print(tag, "sum of kernels: %.1f us per unit" % (tot * 1e3 / n))
    print("   " + "  ".join("%s %.1f" % (r["kernel"], r["total_ms"] * 1e3 / n) for r in rows if r["launches"]))
dev = torch.device("cuda")
W, H = 3840, 2160
P = dict(pyr_scale=0.5, levels=4, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, 9, device=dev)
flows = torch.empty((8, H, W, 2), dtype=torch.float32, device=dev)
ctx = Context(W, H)
for _ in range(4): ctx.farneback_clip(frames, flows, **P)
torch.cuda.synchronize()
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(4): ctx.farneback_clip(frames, flows, **P)
torch.cuda.synchronize()
show(ctx, 32, "4K 5 scales (per frame)")
ctx.close(); del frames, flows; torch.cuda.empty_cache()
W, H = 1920, 1080
ctx = Context(W, H)
import numpy as np
U, V = synth.surf_field(W, H)
f0 = torch.from_numpy(np.stack([U, V], axis=2).astype(np.float32)).to(dev)
ctx.analysis_reset(W, H)
def frame_analysis():
    ctx.streamline_field(f0, 2.0, 1)
    ctx.histogram_accumulate(f0)
    ctx.thresholds()
    ctx.create_flow_accumulate(f0, 40, want=("outmask",))
for _ in range(5): frame_analysis()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): frame_analysis()
torch.cuda.synchronize()
print("analysis wall per frame: %.1f us" % ((time.perf_counter() - t) / 20 * 1e6))
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(10): frame_analysis()
torch.cuda.synchronize()
show(ctx, 10, "1080p per-frame analysis")
