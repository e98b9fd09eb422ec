# streaming state machine: parameters / size changing between pushed frames must not reuse a stale expansion
import sys, numpy as np, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
ctx = Context(640, 480)
clip = synth.surf_clip(320, 240, 4, seed=3)
clip2 = synth.surf_clip(256, 192, 3, seed=4)
P1 = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
variants = {"winsize": dict(P1, winsize=10, flags=256), "poly_n": dict(P1, poly_n=5, poly_sigma=1.1), "levels": dict(P1, levels=3),
            "pyr_scale": dict(P1, pyr_scale=0.7), "iterations": dict(P1, iterations=3), "poly_sigma": dict(P1, poly_sigma=1.5)}
def npy(x): return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
for name, P2 in variants.items():
    ctx.stream_reset()
    ctx.push_frame(clip[0], **P1)
    ctx.push_frame(clip[1], **P1)
    r = ctx.push_frame(clip[2], **P2)                         # changed parameters: primes again, no flow
    got = npy(ctx.push_frame(clip[3], **P2)).copy()           # flow(clip[2] -> clip[3]) with P2
    ref = npy(ctx.calcOpticalFlowFarneback(clip[2], clip[3], None, **P2))
    print("%-10s changed between frames: first call returns %s; next flow identical to a fresh pairwise call: %s" % (name, "no flow" if r is None else "a flow", bool(np.array_equal(got, ref))), flush=True)
# size change mid-stream: the first frame of the new size has no predecessor
ctx.stream_reset()
ctx.push_frame(clip[0], **P1); ctx.push_frame(clip[1], **P1)
r = ctx.push_frame(clip2[0], **P1)
print("size change: first frame of the new size returns", None if r is None else "a flow field", flush=True)
got = npy(ctx.push_frame(clip2[1], **P1)).copy()
ref = npy(ctx.calcOpticalFlowFarneback(clip2[0], clip2[1], None, **P1))
print("size change: next flow identical to pairwise: %s" % bool(np.array_equal(got, ref)), flush=True)
