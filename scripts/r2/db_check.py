"""k_flow_iter2_db (ablate 32) against the production kernel: same bits on several shapes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
ok = True
for (w, h, n, p) in ((640, 480, 5, P), (333, 251, 3, P), (1920, 1080, 9, P), (640, 480, 3, dict(P, flags=256)), (1024, 576, 3, dict(P, levels=4)),
                     (700, 500, 3, dict(P, pyr_scale=0.7, levels=3))):
    clip = torch.as_tensor(synth.surf_clip(w, h, n, seed=3)).cuda()
    with Context(w, h) as ctx:
        ctx.set_option("exact", 0)
        a = ctx.farneback_clip(clip, **p).clone()
        ctx.set_option("ablate", 32)
        b = ctx.farneback_clip(clip, **p).clone()
        torch.cuda.synchronize()
        same = bool(torch.equal(a, b))
        ok &= same
        print(w, h, n, "flags", p["flags"], "levels", p["levels"], "same bits:", same, "" if same else float((a - b).abs().max()), flush=True)
print("ALL SAME" if ok else "MISMATCH", flush=True)
