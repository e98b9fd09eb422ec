# the remaining per-pixel entry points on tiny / degenerate inputs and odd arguments (run under timeout)
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from ripcurrents_amd import RcflowError
from ripcurrents_amd.api import Context
ctx = Context(640, 480)
rng = np.random.RandomState(0)
def npy(x): return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
def run(tag, fn):
    try:
        with np.errstate(all="ignore"):
            print("%-46s %s" % (tag, fn()), flush=True)
    except RcflowError as e:
        print("%-46s error %d" % (tag, e.code), flush=True)
    except Exception as e:
        print("%-46s %s: %s" % (tag, type(e).__name__, str(e)[:70]), flush=True)
for (h, w) in [(1, 1), (1, 7), (9, 1), (2, 2), (3, 5), (10, 10), (11, 13)]:
    f = (rng.randn(h, w, 2) * 1.5).astype(np.float32)
    def t_sub():
        r = f.copy(); orc.subtract_average(r); return bool(np.allclose(npy(ctx.subtructAverage(f.copy())), r, atol=1e-6, equal_nan=True))
    def t_mm():
        r = f.copy(); orc.subtract_mean_magnitude(r); return bool(np.allclose(npy(ctx.subtructMeanMagnitude(f.copy())), r, atol=1e-4, equal_nan=True))
    def t_stab():
        r = f.copy(); orc.stabilizer(r); return bool(np.allclose(npy(ctx.stabilizer(f.copy())), r, atol=1e-6, equal_nan=True))
    def t_edges():
        m = (rng.rand(h, w) < 0.4).astype(np.uint8) * 255
        return bool(np.array_equal(npy(ctx.create_edges(m)), orc.create_edges(m)))
    def t_shear():
        r, _ = orc.shear_rate_to_color(f, 0.0); g, _ = ctx.shearRateToColor(f, 0.0); return bool(np.array_equal(npy(g), r))
    for name, fn in (("subtract_average", t_sub), ("subtract_mean_magnitude", t_mm), ("stabilizer", t_stab), ("create_edges", t_edges), ("shear_rate_to_color", t_shear)):
        run("%dx%d %s" % (w, h, name), fn)
f = (rng.randn(48, 64, 2)).astype(np.float32)
for window in (0, -3, 1, 2, 10 ** 9):
    def t_win():
        avg = np.zeros_like(f); slot = np.zeros_like(f)
        davg, dslot = torch.as_tensor(avg).cuda(), torch.as_tensor(slot).cuda()
        orc.window_mean_update(avg, slot, f, window); ctx.window_mean(davg, dslot, torch.as_tensor(f).cuda(), window)
        return bool(np.array_equal(npy(davg), avg, equal_nan=True))
    run("window_mean window=%d" % window, t_win)
