"""Scale-0 expansion kernel time for a build (RCFLOW_LIB): HIP events over 32-frame launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H, NP = 1920, 1080, 32
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, NP + 1, device=torch.device("cuda"))
flows = torch.empty((NP, H, W, 2), dtype=torch.float32, device="cuda")
with Context(W, H) as ctx:
    for k, v in [a.split("=") for a in sys.argv[1:]]:
        ctx.set_option(k, int(v))
    for _ in range(3): ctx.farneback_clip(frames, flows, **P)
    torch.cuda.synchronize()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(10): ctx.farneback_clip(frames, flows, **P)
    ctx.profile_enable(False)
    rows = {r["kernel"]: r["total_ms"] * 1e3 / r["launches"] for r in ctx.profile_read()}
    print(os.environ.get("RCFLOW_LIB", "default"), " ".join("%s %.1f" % (k, v) for k, v in rows.items() if k.startswith(("polyexp", "flow"))))
