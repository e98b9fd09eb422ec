import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=0, winsize=3, iterations=6, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, 9, device=torch.device("cuda"))
flows = torch.empty((8, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
ctx.set_option("fuse_iters", 0)
def run(tag, **opts):
    for k, v in opts.items(): ctx.set_option(k, v)
    res = []
    for iters in (2, 6):
        P["iterations"] = iters
        for _ in range(2): ctx.farneback_clip(frames, flows, **P)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): ctx.farneback_clip(frames, flows, **P)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t) / 5 / 8)
    per_iter = (res[1] - res[0]) / 4
    print("%-44s %.1f us per pair-iteration (L0, single level)" % (tag, per_iter * 1e6), flush=True)
run("full")
run("no R1 gathers (1)", ablate=1)
run("RA1 gathers only (16)", ablate=16)
run("no window/solve/store (4)", ablate=4)
run("no gathers, no window (5)", ablate=5)
run("empty blocks (8)", ablate=8)
run("full again", ablate=0)
run("full, no remap", xcd_remap=0)
