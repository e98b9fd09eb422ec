# odd parameters and hostile images through the Lucas-Kanade entry point: an error code or the oracle's answer,
# never a crash / hang (run under timeout)
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from ripcurrents_amd import RcflowError, synth
from ripcurrents_amd.api import Context
w, h = 320, 240
ctx = Context(640, 480)
fr = synth.surf_clip(w, h, 2)
rng = np.random.RandomState(2)
pts = np.stack([rng.uniform(5, w - 5, 50), rng.uniform(5, h - 5, 50)], 1).astype(np.float32)
imgs = {"surf": (fr[0], fr[1]), "flat": (np.full((h, w), 9, np.uint8),) * 2,
        "noise": (rng.randint(0, 256, (h, w)).astype(np.uint8), rng.randint(0, 256, (h, w)).astype(np.uint8)),
        "saturated": (np.full((h, w), 255, np.uint8), np.zeros((h, w), np.uint8))}
cases = [dict(), dict(win=(1, 1)), dict(win=(2, 2)), dict(win=(3, 3)), dict(win=(0, 5)), dict(win=(-3, 5)), dict(win=(101, 101)), dict(win=(400, 400)),
         dict(max_level=0), dict(max_level=-1), dict(max_level=10), dict(max_level=100), dict(max_count=0), dict(max_count=-5), dict(max_count=1000),
         dict(epsilon=0.0), dict(epsilon=-1.0), dict(epsilon=float("nan")), dict(epsilon=1e9), dict(crit_type=0), dict(crit_type=1), dict(crit_type=2),
         dict(crit_type=99), dict(flags=8), dict(flags=4), dict(flags=1 << 20), dict(min_eig_threshold=0.0), dict(min_eig_threshold=-1.0),
         dict(min_eig_threshold=1e9), dict(min_eig_threshold=float("nan"))]
for name, (a, b) in imgs.items():
    for kw in cases if name == "surf" else [dict(), dict(win=(50, 50), flags=10, epsilon=0.1)]:
        tag = "%s %s" % (name, kw)
        try:
            with np.errstate(all="ignore"):
                q, st, er = ctx.calcOpticalFlowPyrLK(a, b, pts, **kw)
                q, st = q.cpu().numpy(), st.cpu().numpy()
            try:
                okw = dict(kw)
                rq, rst, _ = orc.pyrlk(a, b, pts, **okw)
                good = (rst == 1) & (st == 1)
                d = float(np.abs(q[good] - rq[good]).max()) if good.any() else 0.0
                print("%-60s ok   status same %s (%d tracked) max diff %.3g" % (tag[:60], bool(np.array_equal(st, rst)), int(st.sum()), d), flush=True)
            except Exception as e:
                print("%-60s ok   (%d tracked); oracle: %s" % (tag[:60], int(st.sum()), str(e)[:50]), flush=True)
        except RcflowError as e:
            print("%-60s error %d" % (tag[:60], e.code), flush=True)
        except Exception as e:
            print("%-60s %s: %s" % (tag[:60], type(e).__name__, str(e)[:60]), flush=True)
