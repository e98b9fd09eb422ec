# The whole per-frame loop of ripcurrents.cpp:194-511 on the device: decoded 1080p BGR frame ->
# 640x480 gray -> flow -> advection -> three display images -> histogram/thresholds -> classify/accumulate
# -> edges -> output frame.  Wall time per frame (one Python call per step, frames resident in HBM).
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context, HistState
dev = torch.device("cuda")
for (sw, sh, w, h) in ((1920, 1080, 640, 480), (1920, 1080, 1920, 1080)):
    T = 12
    gray = synth.surf_clip(sw, sh, T, device=dev)
    bgr = torch.stack([gray, gray, gray], dim=-1).contiguous()
    color = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
    ctx = Context(w, h)
    ctx.analysis_reset(w, h)
    def frame(t, fc):
        f1 = ctx.resize_bgr_to_gray(bgr[t % T], w, h) if (sw, sh) != (w, h) else gray[t % T]
        flow = ctx.push_frame(f1)
        if flow is None: return
        ctx.streamline_field(flow, 2.0, 1)
        for which in (0, 1, 2): ctx.streamline_display(which)
        ctx.histogram_accumulate(flow); ctx.thresholds()
        outs = ctx.create_flow_accumulate(flow, fc, want=("outmask",))
        ctx.create_output(color, ctx.create_edges(outs["outmask"]))
    for t in range(20): frame(t, t + 31)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for t in range(n): frame(t, t + 60)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%dx%d source -> %dx%d loop: %.0f us per frame  %.0f frames/s" % (sw, sh, w, h, dt * 1e6, 1 / dt), flush=True)
    ctx.close()
