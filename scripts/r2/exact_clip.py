"""One 1080p clip through option "exact" for a parameter set (rocprofv3 --kernel-trace target).
usage: exact_clip.py RC215|MAIN264|MAIN1119|AND167 [exact=1] [pairs=8]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
RC215 = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
SETS = dict(RC215=RC215, MAIN264=dict(RC215, flags=256), MAIN1119=dict(RC215, winsize=10, iterations=3, flags=256),
            AND167=dict(RC215, levels=3, winsize=5, iterations=3))
p = SETS[sys.argv[1]]
ex = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
clip = torch.as_tensor(synth.surf_clip(1920, 1080, pairs + 1, seed=1)).cuda()
with Context(1920, 1080) as ctx:
    ctx.set_option("exact", ex)
    for _ in range(3):
        ctx.farneback_clip(clip, **p)
    torch.cuda.synchronize()
