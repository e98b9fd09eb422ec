# per-kernel event times for frame-at-a-time calls (count = 1) at 1080p
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
import sys as _s
for kv in _s.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
for i in range(20): ctx.push_frame(frames[i % 8], flow, **P)
torch.cuda.synchronize()
ctx.profile_enable(True); ctx.profile_reset()
n = 50
for i in range(n): ctx.push_frame(frames[i % 8], flow, **P)
torch.cuda.synchronize()
rows = ctx.profile_read(); ctx.profile_enable(False)
tot = sum(r["total_ms"] for r in rows)
print("sum of kernels per frame: %.1f us" % (tot * 1e3 / n))
print("  " + "  ".join("%s %.1f" % (r["kernel"], r["total_ms"] * 1e3 / n) for r in rows if r["launches"]))
