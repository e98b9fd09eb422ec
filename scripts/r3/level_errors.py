"""Where does the fast path's difference from the oracle come from on a deep pyramid?  (the C3 pixel)

For one 4K pair with five scales: the fast path's and the exact path's (= the oracle's, bit for bit) flow field of every
scale, their difference per scale over the pixels the oracle calls well conditioned along their path, and the worst
final pixel traced back through its ancestors.  Variants: default options, every expansion tap kept (exact_taps)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context, _alias_tensor

W, H = (3840, 2160) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
LEVELS = 4 if len(sys.argv) < 4 else int(sys.argv[3])
P = dict(pyr_scale=0.5, levels=LEVELS, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
clip = synth.surf_clip(W, H, 2, seed=5)
d = torch.as_tensor(clip).cuda()
ref, det_last, det_min, lf = oracle.farneback_diag(clip[0], clip[1], nthreads=8, level_flows=True,
                                                   **{("iters" if k == "iterations" else k): v for k, v in P.items()})
nlev = len(lf)


def level_flows(ctx):
    out = []
    for k in range(nlev):
        p, w, h = C.c_void_p(), C.c_int(), C.c_int()
        ctx._lib.rcflow_debug_level_flow_ptr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        rc = ctx._lib.rcflow_debug_level_flow_ptr(ctx._h, 0, k, 0, C.byref(p), C.byref(w), C.byref(h))
        if rc != 0 or k == 0:
            out.append(None)
            continue
        out.append(_alias_tensor(p.value, w.value * h.value * 2, torch.float32, ctx.device).view(h.value, w.value, 2).cpu().numpy().copy())
    return out


def run(opts):
    with Context(W, H) as ctx:
        for k, v in opts.items():
            ctx.set_option(k, v)
        f = ctx.calcOpticalFlowFarneback(d[0], d[1], None, **P).cpu().numpy()
        ctx.sync()
        return f, level_flows(ctx)


for name, opts in (("exact", dict(exact=1)), ("fast", dict(exact=0)), ("fast+all taps", dict(exact=0, exact_taps=1))):
    f, lv = run(opts)
    err = np.abs(f - ref).max(-1)
    path = det_min > 1e-2
    well = det_last > 1e-2
    print("== %s: final max %.3g | well max %.3g p99.9 %.3g | path max %.3g (share %.3f)" %
          (name, err.max(), err[well].max(), np.percentile(err[well], 99.9), err[path].max(), path.mean()))
    print("   max error over pixels with det_min above: " + "  ".join("%g: %.3g (%.0f %% of px)" % (t, err[det_min > t].max(), 100 * (det_min > t).mean())
                                                                     for t in (1e-2, 2e-2, 5e-2, 1e-1, 1.0)))
    for k in range(nlev - 1, 0, -1):
        if lv[k] is None:
            continue
        e = np.abs(lv[k] - lf[k]).max(-1)
        print("   scale %d (%dx%d): max %.3g  p99.9 %.3g  mean %.3g" % (k, lf[k].shape[1], lf[k].shape[0], e.max(), np.percentile(e, 99.9), e.mean()))
    if name.startswith("fast"):
        e2 = np.where(path, err, 0)
        y, x = np.unravel_index(e2.argmax(), e2.shape)
        print("   worst path-conditioned pixel (x=%d, y=%d): err %.3g det_last %.3g det_min %.3g flow %s ref %s" %
              (x, y, err[y, x], det_last[y, x], det_min[y, x], f[y, x], ref[y, x]))
        for k in range(1, nlev):
            if lv[k] is None:
                continue
            xs, ys = x >> k, y >> k
            y0, y1, x0, x1 = max(ys - 1, 0), ys + 2, max(xs - 1, 0), xs + 2
            e = np.abs(lv[k][y0:y1, x0:x1] - lf[k][y0:y1, x0:x1]).max()
            print("      ancestors at scale %d around (%d, %d): max err %.3g (x %d at scale 0 = %.3g px)" % (k, xs, ys, e, 2 ** k, e * 2 ** k))
