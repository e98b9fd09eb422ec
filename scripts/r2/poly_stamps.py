"""Phase stamps of the scale-0 expansion kernel (diagnostic build with -DRC_STAMPS via RCFLOW_LIB): s_memtime at
0 start, 1 staged (loads arrived + first barrier), 2 fused pyramid done, 3 blur done (before the horizontal pass),
4 horizontal pass done (barrier), 5 vertical pass done, 6 stores issued, 7 stores acknowledged (vmcnt 0)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H, NP = 1920, 1080, 32
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, NP + 1, device=torch.device("cuda"))
flows = torch.empty((NP, H, W, 2), dtype=torch.float32, device="cuda")
with Context(W, H) as ctx:
    for _ in range(3): ctx.farneback_clip(frames, flows, **P)
    ctx.set_option("stamps", 1)
    for _ in range(2): ctx.farneback_clip(frames, flows, **P)
    n = (1020 // 61 + 1) * 8
    buf = (ctypes.c_longlong * n)()
    lib = ctx._lib
    lib.rcflow_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    assert lib.rcflow_debug_read_stamps(ctx._h, buf, n) == 0
s = np.array(buf[:]).reshape(-1, 8)
s = s[(s[:, 0] > 0) & (s[:, 7] > 0)]
d = np.diff(s, axis=1).astype(np.float64)
names = ["wait loads + stage + barrier", "fused pyramid phase", "blur -> tin (+barrier)", "horizontal pass + barrier", "vertical pass", "issue stores", "stores acknowledged"]
print(os.environ.get("RCFLOW_LIB", "default"), "blocks sampled", len(s))
for i, nm in enumerate(names):
    print("  %-32s median %7.0f cycles   p90 %7.0f cycles" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 90)))
print("  %-32s median %7.0f cycles" % ("block total (to stores issued)", np.median(s[:, 6] - s[:, 0])))
