# odd arguments through the C ABI: an error code or a sane result, never a crash / hang
import sys, numpy as np, torch
sys.path.insert(0, '.')
from ripcurrents_amd import RcflowError
from ripcurrents_amd.api import Context
ctx = Context(640, 480)
a = (np.random.RandomState(0).rand(240, 320) * 255).astype(np.uint8)
base = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
cases = [("levels=-1", dict(levels=-1)), ("levels=50", dict(levels=50)), ("winsize=0", dict(winsize=0)), ("winsize=1", dict(winsize=1)),
         ("winsize=2", dict(winsize=2)), ("winsize=-5", dict(winsize=-5)), ("winsize=63", dict(winsize=63)), ("winsize=65", dict(winsize=65)),
         ("winsize=1000", dict(winsize=1000)), ("iterations=0", dict(iterations=0)), ("iterations=-3", dict(iterations=-3)),
         ("iterations=50", dict(iterations=50)), ("poly_n=0", dict(poly_n=0)), ("poly_n=1", dict(poly_n=1)), ("poly_n=-2", dict(poly_n=-2)),
         ("poly_n=32", dict(poly_n=32)), ("poly_n=33", dict(poly_n=33)), ("poly_n=500", dict(poly_n=500)), ("poly_sigma=0", dict(poly_sigma=0.0)),
         ("poly_sigma=-1", dict(poly_sigma=-1.0)), ("poly_sigma=1e-9", dict(poly_sigma=1e-9)), ("poly_sigma=nan", dict(poly_sigma=float("nan"))),
         ("poly_sigma=1e9", dict(poly_sigma=1e9)), ("pyr_scale=0", dict(pyr_scale=0.0)), ("pyr_scale=-0.5", dict(pyr_scale=-0.5)),
         ("pyr_scale=nan", dict(pyr_scale=float("nan"))), ("pyr_scale=0.999", dict(pyr_scale=0.999)), ("pyr_scale=1e-6", dict(pyr_scale=1e-6)),
         ("flags=1<<20", dict(flags=1 << 20)), ("flags=-1", dict(flags=-1))]
for name, kw in cases:
    p = dict(base, **kw)
    try:
        out = ctx.calcOpticalFlowFarneback(a, a[::-1].copy(), None, **p)
        out = out.cpu().numpy() if hasattr(out, "cpu") else np.asarray(out)
        print("%-18s ok   finite=%s max|flow|=%.3g" % (name, bool(np.isfinite(out).all()), float(np.nanmax(np.abs(out)))), flush=True)
    except RcflowError as e:
        print("%-18s error %d %s" % (name, e.code, str(e)[:70]), flush=True)
    except Exception as e:
        print("%-18s %s: %s" % (name, type(e).__name__, str(e)[:80]), flush=True)
