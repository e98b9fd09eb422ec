"""Round-2 probe: option "exact" vs the oracle (bit equality), fast path vs oracle with the
determinant-conditioned metric of SURVEY.md 8(d).  Writes gpurun_out/r2_exact_probe.txt."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context

RC215 = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
SETS = dict(RC215=RC215, MAIN264=dict(RC215, flags=256), MAIN609=dict(RC215, winsize=20, iterations=3, flags=256),
            MAIN1119=dict(RC215, winsize=10, iterations=3, flags=256), AND167=dict(RC215, levels=3, winsize=5, iterations=3))
out = open("gpurun_out/r2_exact_probe.txt", "w")
def P(*a):
    s = " ".join(str(x) for x in a)
    print(s); out.write(s + "\n"); out.flush()

def stats(err, mask=None):
    e = err if mask is None else err[mask]
    if e.size == 0: return "n=0"
    return "n=%d max=%.3g p999=%.3g p99=%.3g frac1e-3=%.5f" % (e.size, e.max(), np.percentile(e, 99.9), np.percentile(e, 99), (e <= 1e-3).mean())

for (w, h) in ((640, 480), (333, 251)):
    clip = synth.surf_clip(w, h, 2, seed=1234)
    with Context(w, h) as ctx:
        # stage: polyexp exact
        I = oracle.pyr_level(clip[0], 0.0, 3, w, h)
        ref = oracle.polyexp(I, 15, 1.2)
        ctx.set_option("exact", 1)
        got = ctx.stage_polyexp(I, 15, 1.2).cpu().numpy()
        ctx.set_option("exact", 0)
        P("polyexp exact %dx%d: equal=%s maxdiff=%g" % (w, h, np.array_equal(got, ref), np.abs(got - ref).max()))
        for name, p in SETS.items():
            o = dict(p); o["iters"] = o.pop("iterations")
            ref, dl, dm = oracle.farneback_diag(clip[0], clip[1], nthreads=8, **o)
            fast = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **p)
            ctx.set_option("exact", 1)
            ex = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **p)
            ctx.set_option("exact", 0)
            e_ex = np.abs(ex - ref).max(-1); e_f = np.abs(fast - ref).max(-1)
            P("%s %dx%d" % (name, w, h))
            P("   exact vs oracle: bit-equal px %.6f  %s" % ((ex == ref).all(-1).mean(), stats(e_ex)))
            P("   fast  vs oracle: all        %s" % stats(e_f))
            P("   fast  det_last>1e-2 (%.3f): %s" % ((dl > 1e-2).mean(), stats(e_f, dl > 1e-2)))
            P("   fast  det_last<=1e-2      : %s" % stats(e_f, dl <= 1e-2))
            P("   fast  det_min>1e-2 (%.3f) : %s" % ((dm > 1e-2).mean(), stats(e_f, dm > 1e-2)))
            if name == "MAIN264" and w == 640:
                np.savez_compressed("gpurun_out/r2_main264_maps.npz", err=e_f.astype(np.float32), det_last=dl.astype(np.float32), det_min=dm, ref=ref, fast=fast)

# cost of the exact path at 1080p
w, h = 1920, 1080
clip = torch.as_tensor(synth.surf_clip(w, h, 9, seed=1)).cuda()
with Context(w, h) as ctx:
    for name in ("RC215", "MAIN264", "MAIN1119"):
        p = SETS[name]
        for ex in (0, 1):
            ctx.set_option("exact", ex)
            ctx.farneback_clip(clip, **p); torch.cuda.synchronize()
            t = time.time()
            for _ in range(3): ctx.farneback_clip(clip, **p)
            torch.cuda.synchronize()
            P("1080p %s exact=%d: %.1f us per pair" % (name, ex, (time.time() - t) / 24 * 1e6))
