# per-kernel event times for option sets given as "name=value,name=value" arguments
import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, 17, device=torch.device("cuda"))
flows = torch.empty((16, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
for arg in sys.argv[1:]:
    for kv in arg.split(","):
        k, v = kv.split("="); ctx.set_option(k, int(v))
    for _ in range(3): ctx.farneback_clip(frames, flows, **P)
    torch.cuda.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(6): ctx.farneback_clip(frames, flows, **P)
    torch.cuda.synchronize()
    rows = ctx.profile_read(); ctx.profile_enable(False)
    tot = sum(r["total_ms"] for r in rows)
    print(arg, "sum of kernels per frame: %.1f us" % (tot * 1e3 / 6 / 16))
    print("   " + "  ".join("%s %.1f" % (r["kernel"], r["total_ms"] * 1e3 / 6 / 16) for r in rows if r["launches"]))
