# adversarial inputs (edges of thresholds, zeros, denormals, huge, Inf, NaN) through the per-pixel analysis
# kernels against the oracle: classification / accumulation, dense advection, colour coding
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from ripcurrents_amd.api import Context, HistState
w, h = 512, 256
n = w * h
rng = np.random.RandomState(5)
ex = np.array([0.0, -0.0, 1e-45, 1e-39, 1e-20, 0.2, 0.5, 0.05, 1.0, 1.7, 2.5, 1e10, 3e38, np.inf, -np.inf, np.nan], np.float32)
f = (rng.randn(n, 2) * 0.8).astype(np.float32)
idx = rng.rand(n) < 0.3
f[idx, 0] = ex[rng.randint(0, len(ex), idx.sum())]
idx = rng.rand(n) < 0.3
f[idx, 1] = ex[rng.randint(0, len(ex), idx.sum())]
# magnitudes exactly on the classification thresholds 0.2 / 0.5
k = rng.rand(n) < 0.1
th = rng.rand(k.sum()) * 2 * np.pi
r = np.where(rng.rand(k.sum()) < 0.5, 0.2, 0.5)
f[k, 0] = (r * np.cos(th)).astype(np.float32); f[k, 1] = (r * np.sin(th)).astype(np.float32)
f = f.reshape(h, w, 2)
ctx = Context(w, h)
def same(a, b): return bool(np.array_equal(a, b, equal_nan=True))
with np.errstate(all="ignore"):
    ctx.analysis_reset(w, h)
    st, ost = HistState(), orc.HistState()
    ctx.create_histogram(f, st)
    polar = orc.flow_to_polar(f)
    orc.create_histogram(polar, ost)
    print("histogram", same(st.hist2d, ost.hist2d), st.UPPER, ost.UPPER, same(st.UPPER2d, ost.UPPER2d))
    acc = np.zeros((h, w, 3), np.float32)
    for fc in (1, 31, 32, 40):
        outs = ctx.create_flow_accumulate(f, fc)
        wc = np.zeros((h, w, 3), np.float32); acc2 = np.zeros((h, w, 3), np.float32)
        p2 = polar.copy()
        orc.create_flow(p2, wc, acc2, ost.UPPER, 0.5, 0.2, ost.UPPER2d)
        out = np.zeros((h, w, 3), np.float32); mask = np.zeros((h, w), np.uint8)
        orc.create_accumulationbuffer(acc, acc2, out, mask, fc)
        print("classify fc=%d" % fc, same(outs["waterclass"].cpu().numpy(), wc), same(outs["polar"].cpu().numpy(), p2),
              same(outs["out"].cpu().numpy(), out), same(outs["outmask"].cpu().numpy(), mask), same(ctx.accumulator(w, h), acc[..., 0]))
    ctx.analysis_reset(w, h)
    pt = np.zeros((h, w, 2), np.float32); dist = np.zeros((h, w), np.float32)
    for t in range(3):
        ctx.streamline_field(f, 2.0, 2, UPPER=1.7)
        orc.streamline_field(pt, dist, f, 2.0, 2, 1.7)
    gpt, gdist = ctx.streamline_field_state(w, h)
    print("advect field", same(gpt, pt), same(gdist, dist), "mismatching px", int((~np.isclose(gpt, pt, equal_nan=True, rtol=0, atol=0)).any(-1).sum()))
    pts = (rng.rand(300, 2) * [w, h]).astype(np.float32)
    for variant in (0, 1, 2, 3, 4):
        g, _ = ctx.streamline(pts.copy(), f, 0.1, 20, 1.7, variant=variant)
        o = pts.copy(); orc.streamline_points(o, f, 0.1, 20, 1.7, variant=variant)
        print("streamline variant", variant, same(g.cpu().numpy() if hasattr(g, "cpu") else np.asarray(g), np.asarray(o)))
