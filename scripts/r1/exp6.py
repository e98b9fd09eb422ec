import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H, S = 1920, 1080, 16
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = [synth.surf_clip(W, H, 1, seed=s, t0=t, device=torch.device("cuda"))[0] for t in range(2) for s in range(S)]
fa = torch.stack(frames[:S]); fb = torch.stack(frames[S:])
stage = torch.empty_like(fa)
flows = torch.empty((S, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for g in (False, True):
        ctx.batch_reset()
        def step(i):
            stage.copy_(fa if i % 2 == 0 else fb)
            ctx.push_batch(stage, flows, use_graph=g, **P)
        for i in range(6): step(i)
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 20
        for i in range(n): step(i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
        print("16 streams lockstep, graph=%s: %.1f us/frame  %.0f fps" % (g, dt / S * 1e6, S / dt), flush=True)
