# sweep kernel vs tile kernel for the large Gaussian windows: bit identity + time
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
variants = [("tile", 65536), ("sweep", 0)] + [(a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:]]
for (W, H) in [(1920, 1080)]:
    ctx = Context(W, H)
    for npairs in (16,):
        frames = synth.surf_clip(W, H, npairs + 1, device=torch.device("cuda"))
        for ws in (10, 20):
            P = dict(pyr_scale=0.5, levels=2, winsize=ws, iterations=3, poly_n=15, poly_sigma=1.2, flags=256)
            out = {}
            for name, abl in variants:
                ctx.set_option("ablate", abl)
                flows = torch.zeros((npairs, H, W, 2), dtype=torch.float32, device="cuda")
                ctx.farneback_clip(frames, flows, **P)
                torch.cuda.synchronize()
                reps = max(2, 32 // npairs)
                t0 = time.perf_counter()
                for _ in range(reps): ctx.farneback_clip(frames, flows, **P)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / (reps * npairs)
                out[name] = (flows.cpu().numpy(), dt)
            ref = out["tile"][0]
            print("%dx%d x%d winsize %d:" % (W, H, npairs, ws), "  ".join("%s %.1f us/pair same=%s" % (k, v[1] * 1e6, bool((v[0] == ref).all())) for k, v in out.items()), flush=True)
