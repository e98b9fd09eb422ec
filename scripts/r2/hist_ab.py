"""Histogram kernel A/B on 32 real 1080p flow fields: first form (ablate bit 16777216) vs the row-column form."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H, NP = 1920, 1080, 32
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, NP + 1, device=torch.device("cuda"))
flows = torch.empty((NP, H, W, 2), dtype=torch.float32, device="cuda")
noise = torch.randn((NP, H, W, 2), dtype=torch.float32, device="cuda") * 0.8
with Context(W, H) as ctx:
    ctx.farneback_clip(frames, flows, **P)
    ctx.analysis_reset(W, H)
    res = {}
    for name, fl in (("surf flow", flows), ("gaussian noise", noise)):
        for abl in (16777216, 0, 16777216, 0):
            ctx.set_option("ablate", abl)
            ctx.histogram_reset()
            ctx.histogram_accumulate_clip(fl); torch.cuda.synchronize()
            words = ctx.histogram_words().cpu().numpy().copy()
            res.setdefault(name, []).append(words)
            ctx.histogram_reset()
            t0 = time.perf_counter()
            for _ in range(20): 
                ctx.histogram_reset(); ctx.histogram_accumulate_clip(fl)
            torch.cuda.synchronize()
            print("%s  ablate=%d: %.1f us per 32 fields" % (name, abl, (time.perf_counter() - t0) / 20 * 1e6))
        assert all(np.array_equal(res[name][0], r) for r in res[name]), "counts differ between the two forms"
    print("counts identical")
