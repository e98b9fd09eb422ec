import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, 9, device=torch.device("cuda"))
flows = torch.empty((8, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
for _ in range(3): ctx.farneback_clip(frames, flows, **P)
ctx.set_option("stamps", 1)
for _ in range(2): ctx.farneback_clip(frames, flows, **P)
n = (4080 // 61 + 1) * 8
buf = (ctypes.c_longlong * n)()
lib = ctx._lib
lib.rcflow_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert lib.rcflow_debug_read_stamps(ctx._h, buf, n) == 0
s = np.array(buf[:]).reshape(-1, 8)
s = s[s[:, 0] > 0]
d = np.diff(s[:, :7], axis=1).astype(np.float64)
names = ["issue loads", "wait loads+LDS store+sync", "stage A (gather,M0,sync)", "stage B/C (win,solve,gather,M1)", "M1 write+sync", "stage D (win,solve,store)"]
print("blocks sampled", len(s), " (s_memtime ticks = 100 MHz => 10 ns each)")
for i, nm in enumerate(names):
    print("%-34s median %7.0f ns   p90 %7.0f ns" % (nm, np.median(d[:, i]) * 10, np.percentile(d[:, i], 90) * 10))
print("%-34s median %7.0f ns" % ("block total", np.median(s[:, 6] - s[:, 0]) * 10))
