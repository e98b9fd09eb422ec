import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
ctx = Context(W, H)
frames = synth.surf_clip(W, H, 9, device=torch.device("cuda"))
flows = torch.empty((8, H, W, 2), dtype=torch.float32, device="cuda")
p = dict(pyr_scale=0.5, levels=2, poly_n=15, poly_sigma=1.2, winsize=20, iterations=3, flags=256)
for ab in (0, 16384, 32768, 49152, 0):
    ctx.set_option("ablate", ab)
    for _ in range(2): ctx.farneback_clip(frames, flows, **p)
    torch.cuda.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(3): ctx.farneback_clip(frames, flows, **p)
    torch.cuda.synchronize()
    rows = ctx.profile_read(); ctx.profile_enable(False)
    print("ablate=%-6d " % ab + "  ".join("%s %.1f" % (r["kernel"], r["total_ms"] * 1e3 / 3 / 8) for r in rows if r["launches"] and "iter" in r["kernel"]), flush=True)
