# polyexp tile-height variants: parity of the whole path between them + timing
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
for th in (32, 48, 32, 48):
    ctx.set_option("poly_tile_h", th)
    best = 1e9
    for rep in range(3):
        for _ in range(2): ctx.farneback_clip(frames, flows, **P)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): ctx.farneback_clip(frames, flows, **P)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 5 / 16)
    out = flows.cpu().numpy().copy()
    if ref is None: ref = out
    print("poly_tile_h=%d  %.1f us/frame  %.0f fps   identical to first: %s" % (th, best * 1e6, 1 / best, np.array_equal(out, ref)), flush=True)
