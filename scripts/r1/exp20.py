# per-kernel event times for a parameter set: python scripts/exp20.py winsize iterations flags [levels] [opt=val,...]
import sys, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
ws, it, fl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lv = int(sys.argv[4]) if len(sys.argv) > 4 else 2
P = dict(pyr_scale=0.5, levels=lv, winsize=ws, iterations=it, poly_n=15, poly_sigma=1.2, flags=fl)
frames = synth.surf_clip(W, H, 17, device=torch.device("cuda"))
flows = torch.empty((16, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
for arg in (sys.argv[5:] or ["ablate=0"]):
    for kv in arg.split(","):
        k, v = kv.split("="); ctx.set_option(k, int(v))
    for _ in range(2): ctx.farneback_clip(frames, flows, **P)
    torch.cuda.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(4): ctx.farneback_clip(frames, flows, **P)
    torch.cuda.synchronize()
    rows = ctx.profile_read(); ctx.profile_enable(False)
    tot = sum(r["total_ms"] for r in rows)
    print(arg, "winsize %d iters %d flags %d: sum of kernels per frame: %.1f us" % (ws, it, fl, tot * 1e3 / 4 / 16))
    print("   " + "  ".join("%s %.1f" % (r["kernel"], r["total_ms"] * 1e3 / 4 / 16) for r in rows if r["launches"]))
