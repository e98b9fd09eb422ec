# which inputs make the GPU histogram differ from the oracle's: one field per category
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from ripcurrents_amd.api import Context, HistState
ctx = Context(2048, 1024)
def cmp(name, f):
    f = np.ascontiguousarray(f.reshape(-1, 64, 2).astype(np.float32))
    h, w = f.shape[:2]
    st, ost = HistState(), orc.HistState()
    ctx.analysis_reset(w, h)
    with np.errstate(all="ignore"):
        ctx.create_histogram(f, st)
        orc.create_histogram(orc.flow_to_polar(f), ost)
    d = (st.hist2d.astype(np.int64) - ost.hist2d.astype(np.int64))
    print("%-28s n=%d gpu sum %d oracle sum %d, bins differing %d, max |diff| %d" % (name, w * h, st.histsum, ost.histsum.value, int((d != 0).sum()), int(np.abs(d).max())), flush=True)
    return d
rng = np.random.RandomState(11)
q = 1 << 16
k = rng.randint(1, 51, q).astype(np.float64) / 20.0; th = rng.rand(q) * 2 * np.pi
cmp("mag edges", np.stack([k * np.cos(th), k * np.sin(th)], -1))
m = rng.randint(0, 37, q) * (np.pi / 18); r = rng.rand(q) * 2.6
f = np.stack([r * np.cos(m), r * np.sin(m)], -1).astype(np.float32)
f = (f.view(np.int32) + rng.randint(-3, 4, (q, 2)).astype(np.int32)).view(np.float32)
cmp("angle edges nudged", f)
v = (rng.randint(-60, 61, (q, 2)) * 0.05).astype(np.float32)
cmp("lattice", v)
cmp("randn", rng.randn(q, 2) * 0.8)
ex = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39, 1e-30, 1e-20, 1e-10, 2.45, 2.5, 2.55, 1e10, 1e20, 3e38, np.inf, -np.inf, np.nan, 0.05, 0.1, 1.0], np.float32)
for a in ex:
    g = np.stack([np.full(len(ex) * 64, a, np.float32), np.repeat(ex, 64)], -1)
    d = cmp("x = %r vs all y" % float(a), g)
