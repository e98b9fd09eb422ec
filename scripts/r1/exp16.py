# frame-at-a-time modes at 1080p: push_frame (eager), push_batch with 1/2/4 streams eager vs hipGraph
import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
dev = torch.device("cuda")
frames = synth.surf_clip(W, H, 8, device=dev)
ctx = Context(W, H)
flow = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    ctx.stream_reset() if hasattr(ctx, "stream_reset") else None
    for i in range(10): ctx.push_frame(frames[i % 8], flow, **P)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 200
    for i in range(n): ctx.push_frame(frames[i % 8], flow, **P)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print("push_frame eager:            %.1f us/frame  %.0f fps" % (dt * 1e6, 1 / dt), flush=True)
    for S in (1, 2, 4):
        stage = torch.empty((S, H, W), dtype=torch.uint8, device=dev)
        flows = torch.empty((S, H, W, 2), dtype=torch.float32, device=dev)
        for g in (False, True):
            ctx.batch_reset()
            def step(i):
                stage.copy_(frames[(i % 4) * 1:(i % 4) * 1 + S] if S <= 4 else frames[:S])
                ctx.push_batch(stage, flows, use_graph=g, **P)
            for i in range(10): step(i)
            torch.cuda.synchronize(); t = time.perf_counter()
            n = 100
            for i in range(n): step(i)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n / S
            print("push_batch S=%d graph=%-5s:   %.1f us/frame  %.0f fps" % (S, g, dt * 1e6, 1 / dt), flush=True)
