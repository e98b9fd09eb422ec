import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=0, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, 9, device=torch.device("cuda"))
flows = torch.empty((8, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
def run(tag, **opts):
    for k, v in opts.items(): ctx.set_option(k, v)
    res = []
    for iters in (2, 6):     # single level: the difference isolates 2 fused launches
        P["iterations"] = iters
        best = 1e9
        for rep in range(3):
            for _ in range(2): ctx.farneback_clip(frames, flows, **P)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(5): ctx.farneback_clip(frames, flows, **P)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 5 / 8)
        res.append(best)
    print("%-44s %.1f us per pair per fused launch (scale 0, mode 1)" % (tag, (res[1] - res[0]) / 2 * 1e6), flush=True)
run("full fused")
run("stage A only (1)", ablate=1)
run("no second gather (2)", ablate=2)
run("no final window/solve/store (4)", ablate=4)
run("no 2nd gather, no final (6)", ablate=6)
run("full again", ablate=0)
