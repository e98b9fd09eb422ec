# histogram kernel: time per 32-field launch for hist_blocks values (and per-kernel events)
import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H, NP = 1920, 1080, 32
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, NP + 1, device=torch.device("cuda"))
flows = torch.empty((NP, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
ctx.analysis_reset(W, H)
ctx.farneback_clip(frames, flows, **P)
torch.cuda.synchronize()
for hb in [int(a) for a in sys.argv[1:]] or [0]:
    ctx.set_option("hist_blocks", hb)
    for _ in range(3): ctx.histogram_accumulate_clip(flows)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): ctx.histogram_accumulate_clip(flows)
    torch.cuda.synchronize()
    print("hist_blocks %6d: %.1f us per 32 fields" % (hb, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
ctx.set_option("hist_blocks", 0)
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(10): ctx.histogram_accumulate_clip(flows)
torch.cuda.synchronize()
rows = ctx.profile_read(); ctx.profile_enable(False)
print("  ".join("%s %.1f us" % (r["kernel"], r["total_ms"] * 1e3 / max(r["launches"], 1)) for r in rows if r["launches"]))
