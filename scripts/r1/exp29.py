# adversarial inputs through the colour / display / delta kernels against the oracle
import sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from oracle import oracle as orc
from ripcurrents_amd.api import Context, HistState
from test_gpu_analysis import _adversarial_flow
w, h = 512, 256
f = _adversarial_flow(w, h, 7)
ctx = Context(w, h)
eq = lambda a, b: bool(np.array_equal(a, b, equal_nan=True))
def npy(x): return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
with np.errstate(all="ignore"):
    pt = np.zeros((h, w, 2), np.float32); ref = pt.copy()
    for _ in range(3): orc.get_delta_field(ref, f, 2.0, 1.8)
    d = torch.as_tensor(pt).cuda()
    for _ in range(3): d = ctx.get_delta_field(d, f, 2.0, 1.8)
    print("get_delta_field", eq(npy(d), ref))
    md = gmd = 0.0
    for t in range(2):
        r, md = orc.vector_to_color(f, md); g, gmd = ctx.vectorToColor(f, gmd); g = npy(g)
        dh = np.abs(g[..., 0].astype(int) - r[..., 0].astype(int))
        print("vector_to_color", t, "max", gmd, md, "sat/val equal", eq(g[..., 1:], r[..., 1:]), "hue diff frac", float((dh > 0).mean()), int(dh.max()))
    mf = gmf = 0.0
    for t in range(2):
        r, mf = orc.shear_rate_to_color(f, mf); g, gmf = ctx.shearRateToColor(f, gmf)
        print("shear_rate_to_color", t, gmf, mf, eq(npy(g), r), int((npy(g) != r).any(-1).sum()))
    hsv = np.stack([f[..., 0] * 100, np.abs(f[..., 1]), f[..., 0]], -1).astype(np.float32)
    print("hsv_to_bgr", eq(npy(ctx.hsv_to_bgr(hsv)), orc.hsv_to_bgr(hsv)), int((~np.isclose(npy(ctx.hsv_to_bgr(hsv)), orc.hsv_to_bgr(hsv), equal_nan=True, rtol=0, atol=0)).any(-1).sum()))
    ctx.analysis_reset(w, h)
    for _ in range(3): ctx.streamline_field(f, 2.0, 2, UPPER=1e30)
    spt, sdist = ctx.streamline_field_state(w, h)
    print("state has nan/inf:", bool(~np.isfinite(spt).all()), bool(~np.isfinite(sdist).all()))
    for which in (0, 1, 2):
        img, mx = ctx.streamline_display(which); r, rmx = orc.streamline_display(spt, sdist, which)
        print("display", which, "max", mx, rmx, eq(npy(img), r), int((npy(img) != r).any(-1).sum()))
    print("positions", eq(npy(ctx.streamline_positions()), orc.streamline_positions(spt)))
