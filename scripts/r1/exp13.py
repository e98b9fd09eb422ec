# large-window variants: new (default) vs earlier generic kernel (ablate 8192), same process
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
dev = torch.device("cuda")
W, H = 1920, 1080
ctx = Context(W, H)
frames = synth.surf_clip(W, H, 9, device=dev)
flows = torch.empty((8, H, W, 2), dtype=torch.float32, device=dev)
base = dict(pyr_scale=0.5, levels=2, poly_n=15, poly_sigma=1.2)
for name, p in (("gauss win10 it3", dict(base, winsize=10, iterations=3, flags=256)),
                ("gauss win20 it3", dict(base, winsize=20, iterations=3, flags=256)),
                ("box win5 it3 4 scales", dict(base, levels=3, winsize=5, iterations=3, flags=0)),
                ("box win10 it3", dict(base, winsize=10, iterations=3, flags=0))):
    res = {}
    for ab in (8192, 0, 32768, 0, 32768):
        ctx.set_option("ablate", ab)
        for _ in range(2): ctx.farneback_clip(frames, flows, **p)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): ctx.farneback_clip(frames, flows, **p)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3 / 8
        res.setdefault(ab, []).append(dt * 1e6)
        out = flows.cpu().numpy().copy()
        if ab == 8192: ref = out
        else: same = np.array_equal(out, ref)
    print("%-24s generic %.0f us  32x32 %.0f us  default %.0f us/frame (%.0f fps)  identical: %s" % (name, min(res[8192]), min(res[32768]), min(res[0]), 1e6 / min(res[0]), same), flush=True)
