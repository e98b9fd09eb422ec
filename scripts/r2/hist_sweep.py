"""Histogram kernel: us per 32 fields over the block cap (option hist_blocks), both kernel forms, HIP-event kernel time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H, NP = 1920, 1080, 32
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, NP + 1, device=torch.device("cuda"))
flows = torch.empty((NP, H, W, 2), dtype=torch.float32, device="cuda")
with Context(W, H) as ctx:
    ctx.farneback_clip(frames, flows, **P)
    ctx.analysis_reset(W, H)
    for abl in (16777216, 0):
        for hb in ((1024, 1280, 2560, 3840, 5120, 7680, 10240, 16384) if not abl else (16384,)):
            ctx.set_option("ablate", abl); ctx.set_option("hist_blocks", hb)
            for _ in range(3):
                ctx.histogram_reset(); ctx.histogram_accumulate_clip(flows)
            torch.cuda.synchronize()
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(10):
                ctx.histogram_reset(); ctx.histogram_accumulate_clip(flows)
            ctx.profile_enable(False)
            r = [x for x in ctx.profile_read() if x["kernel"].startswith("polar_hist")][0]
            print("form %s hist_blocks %5d: %.1f us (hist + fold)" % ("v1" if abl else "rows", hb, r["total_ms"] * 1e3 / r["launches"]), flush=True)
