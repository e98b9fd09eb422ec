import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as orc
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
ctx = Context(1920, 1080)
for (w, h, k) in [(640, 480, 2), (256, 64, 2), (640,480,3)]:
    img = synth.surf_clip(w, h, 1, seed=7)[0]
    g = orc.level_geometry(w, h, 0.5, 8, k)
    ref = orc.pyr_level(img, g["sigma"], g["ksize"], g["w"], g["h"])
    got = ctx.stage_pyr_level(img, 0.5, k).cpu().numpy()
    d = np.abs(got - ref)
    print(w, h, k, g, "max", d.max(), "exact frac", (d == 0).mean())
    print(" ref[0,:6]", ref[0, :6]); print(" got[0,:6]", got[0, :6])
    # test hypotheses: blur only horizontally / vertically
    f = img.astype(np.float32)
    kk = orc.gaussian_kernel(g["ksize"], g["sigma"])
    print(" kernel", kk)
