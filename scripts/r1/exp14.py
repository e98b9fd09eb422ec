# two-stream overlap of the clip path: option overlap=0/1, 33- and 65-frame clips
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
dev = torch.device("cuda")
for T in (33, 65):
    frames = synth.surf_clip(W, H, T, device=dev)
    flows = torch.empty((T - 1, H, W, 2), dtype=torch.float32, device=dev)
    ctx = Context(W, H)
    ctx.analysis_reset(W, H)
    ref = None
    for ov in (0, 1, 0, 1, 0, 1):
        ctx.set_option("overlap", ov)
        def step():
            ctx.farneback_clip(frames, flows, **P)
            ctx.histogram_accumulate_clip(flows)
            ctx.thresholds()
        for _ in range(4): step()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10 / (T - 1)
        out = flows[::7].cpu().numpy().copy()
        if ref is None: ref = out
        print("T=%d overlap=%d  %.1f us/frame  %.0f fps  identical: %s" % (T, ov, dt * 1e6, 1 / dt, np.array_equal(out, ref)), flush=True)
    ctx.close(); del frames, flows; torch.cuda.empty_cache()
