# flow tile variants (ablate bits) on the 1080p clip: equality with the default + per-kernel time
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, 17, device=torch.device("cuda"))
flows = torch.empty((16, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
ref = None
for ab in [int(a) for a in sys.argv[1:]]:
    ctx.set_option("ablate", ab)
    for _ in range(3): ctx.farneback_clip(frames, flows, **P)
    torch.cuda.synchronize()
    out = flows[::5].cpu().numpy().copy()
    if ref is None: ref = out
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(6): ctx.farneback_clip(frames, flows, **P)
    torch.cuda.synchronize()
    rows = ctx.profile_read(); ctx.profile_enable(False)
    d = {r["kernel"]: r["total_ms"] * 1e3 / 6 / 16 for r in rows if r["launches"]}
    print("ablate=%-5d same bits: %-5s flow@0 %.1f  @1 %.1f  @2 %.1f us/frame" % (ab, np.array_equal(out, ref), d["flow_iter_x2@0"], d["flow_iter_x2@1"], d["flow_iter_x2@2"]), flush=True)
