"""Cost of option "exact" on 1080p clips: us per pair for RC215 / MAIN264 / MAIN1119 / AND167 at 8 and 32 pairs."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
RC215 = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
SETS = dict(RC215=RC215, MAIN264=dict(RC215, flags=256), MAIN1119=dict(RC215, winsize=10, iterations=3, flags=256),
            AND167=dict(RC215, levels=3, winsize=5, iterations=3))
names = sys.argv[1:] or ["RC215", "AND167", "MAIN264"]
w, h = 1920, 1080
for npairs in (8, 32):
    clip = torch.as_tensor(synth.surf_clip(w, h, npairs + 1, seed=1)).cuda()
    out = torch.empty((npairs, h, w, 2), dtype=torch.float32, device="cuda")
    with Context(w, h) as ctx:
        for name in names:
            for ex in (0, 1):
                ctx.set_option("exact", ex)
                for _ in range(2): ctx.farneback_clip(clip, out, **SETS[name])
                torch.cuda.synchronize()
                t = time.perf_counter()
                reps = 5
                for _ in range(reps): ctx.farneback_clip(clip, out, **SETS[name])
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t) / reps / npairs
                print("%-8s pairs %2d exact %d: %7.1f us per pair  checksum %r" % (name, npairs, ex, dt * 1e6, float(out.double().sum())), flush=True)
