# 4K five-scale per-kernel times (BASELINE config 3), 8 pairs per clip
import sys, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H, NP = 3840, 2160, 8
P = dict(pyr_scale=0.5, levels=4, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, NP + 1, device=torch.device("cuda"))
flows = torch.empty((NP, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
for _ in range(3): ctx.farneback_clip(frames, flows, **P)
torch.cuda.synchronize()
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(5): ctx.farneback_clip(frames, flows, **P)
torch.cuda.synchronize()
rows = ctx.profile_read(); ctx.profile_enable(False)
tot = sum(r["total_ms"] for r in rows)
print("sum of kernels per frame: %.1f us" % (tot * 1e3 / 5 / NP))
print("  " + "  ".join("%s %.1f" % (r["kernel"], r["total_ms"] * 1e3 / 5 / NP) for r in rows if r["launches"]))
