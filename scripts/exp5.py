import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
T = 33
frames = synth.surf_clip(W, H, T, device=torch.device("cuda"))
flows = torch.empty((T - 1, H, W, 2), dtype=torch.float32, device="cuda")
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = Context(W, H, streams=ns)
for s in range(ns): ctx.use_own_stream(s)
per = (T - 1) // ns
def step():
    for s in range(ns):
        a = s * per
        ctx.farneback_clip(frames[a:a + per + 1], flows[a:a + per], stream=s, **P)
    for s in range(ns): ctx.sync(s)
best = 1e9
for rep in range(3):
    for _ in range(2): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 5 / (per * ns))
print("streams=%d  %.1f us/frame  %.0f fps" % (ns, best * 1e6, 1 / best), flush=True)
