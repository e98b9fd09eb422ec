# fused pyramid (option fuse_pyr) vs separate pyramid launches: bit identity + per-kernel times
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
for (W, H, lv, n) in [(1920, 1080, 2, 17), (640, 480, 2, 5), (332, 252, 2, 4), (328, 244, 1, 4), (1024, 512, 4, 4), (64, 32, 2, 4), (8, 8, 1, 4), (72, 40, 2, 4)]:
    ctx = Context(W, H)
    frames = synth.surf_clip(W, H, n, device=torch.device("cuda"), seed=3)
    p = dict(P, levels=lv)
    out = {}
    for fz in (0, 1):
        ctx.set_option("fuse_pyr", fz)
        flows = torch.zeros((n - 1, H, W, 2), dtype=torch.float32, device="cuda")
        ctx.farneback_clip(frames, flows, **p)
        torch.cuda.synchronize()
        out[fz] = flows.cpu().numpy()
        if W == 1920:
            for _ in range(3): ctx.farneback_clip(frames, flows, **p)
            torch.cuda.synchronize()
            ctx.profile_enable(True); ctx.profile_reset()
            for _ in range(6): ctx.farneback_clip(frames, flows, **p)
            torch.cuda.synchronize()
            rows = ctx.profile_read(); ctx.profile_enable(False)
            tot = sum(r["total_ms"] for r in rows)
            print("fuse_pyr=%d sum of kernels per frame: %.1f us  " % (fz, tot * 1e3 / 6 / 16) + "  ".join("%s %.1f" % (r["kernel"], r["total_ms"] * 1e3 / 6 / 16) for r in rows if r["launches"]))
    print("%dx%d levels %d: identical=%s finite=%s" % (W, H, lv, bool((out[0] == out[1]).all()), bool(np.isfinite(out[1]).all())), flush=True)
