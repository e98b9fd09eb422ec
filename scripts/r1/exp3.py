import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, 17, device=torch.device("cuda"))
flows = torch.empty((16, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
def run(tag, **opts):
    for k, v in opts.items(): ctx.set_option(k, v)
    best = 1e9
    for rep in range(3):
        for _ in range(2): ctx.farneback_clip(frames, flows, **P)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): ctx.farneback_clip(frames, flows, **P)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 5 / 16)
    print("%-40s %.1f us/frame  %.0f fps" % (tag, best * 1e6, 1 / best), flush=True)
for a in sys.argv[1:]:
    k, v = a.split("=")
    run(a, **{k: int(v)})
