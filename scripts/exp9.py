import sys, torch, numpy as np
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 640, 480
P = dict(pyr_scale=0.5, levels=0, winsize=3, iterations=1, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.translating_clip(W, H, 2, device=torch.device("cuda"))
ctx = Context(W, H)
outs = []
for ab in (2048, 0):
    ctx.set_option("ablate", ab)
    f = ctx.calcOpticalFlowFarneback(frames[0], frames[1], None, **P)
    torch.cuda.synchronize()
    outs.append(f.cpu().numpy().copy())
d = np.abs(outs[0] - outs[1]).max(axis=2)
print("max diff", d.max(), "frac>1e-6", (d > 1e-6).mean())
ys, xs = np.nonzero(d > 1e-6)
if len(ys):
    print("rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
    colmask = (d > 1e-6).any(axis=0); rowmask = (d > 1e-6).any(axis=1)
    print("cols with diff:", "".join("X" if colmask[i:i+8].any() else "." for i in range(0, W, 8)))
    print("rows with diff:", "".join("X" if rowmask[i:i+8].any() else "." for i in range(0, H, 8)))
    print(d[200:204, 300:308])
sys.path.insert(0, 'oracle')
import oracle as orc
fr = frames.cpu().numpy()
ref = orc.farneback(fr[0], fr[1], pyr_scale=0.5, levels=0, winsize=3, iters=1, poly_n=15, poly_sigma=1.2, flags=0)
for name, o in zip(("slow(ablate 2048)", "fast"), outs):
    e = np.abs(o - ref).max(axis=2)
    print(name, "vs oracle: max", e.max(), "frac>1e-3", (e > 1e-3).mean())
