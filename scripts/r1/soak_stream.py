# long-run soak of the frame-at-a-time path: 1500 frames through push_frame + histogram + thresholds +
# classify/accumulate + advection, twice; the two runs must agree bit for bit and device memory must not grow
import sys, torch, numpy as np
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context, HistState
W, H, N = 640, 480, 1500
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
dev = torch.device("cuda")
frames = synth.surf_clip(W, H, 32, device=dev)

def run():
    ctx = Context(W, H)
    ctx.analysis_reset(W, H)
    flow = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
    mem0 = None
    for i in range(N):
        ctx.push_frame(frames[i % 32], flow, **P)
        if i == 0:
            continue
        ctx.histogram_accumulate(flow)
        ctx.thresholds()
        ctx.create_flow_accumulate(flow, i, want=())
        ctx.streamline_field(flow, 2.0, 1)
        if i == 100:
            torch.cuda.synchronize(); mem0 = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    mem1 = torch.cuda.mem_get_info()[0]
    st = HistState(); ctx.histogram_read(st)
    acc = ctx.accumulator(W, H).copy()
    pt, dist = ctx.streamline_field_state(W, H)
    assert np.isfinite(flow.cpu().numpy()).all()
    return st.hist2d.copy(), float(st.UPPER), acc, pt.copy(), dist.copy(), mem0 - mem1

a = run(); b = run()
same = all(np.array_equal(x, y) for x, y in zip(a[:5], b[:5]))
print("frames", N, "runs identical:", same, "histsum", int(a[0].sum()), "UPPER", a[1], "free-memory drift (bytes):", a[5], b[5])
