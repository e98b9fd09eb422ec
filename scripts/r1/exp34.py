# resize + gray on odd size pairs (upscale, 1-pixel sources, huge factors, non-integer factors): an error
# code or the oracle's image, never a crash (run under timeout)
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from ripcurrents_amd import RcflowError
from ripcurrents_amd.api import Context
ctx = Context(1920, 1080)
rng = np.random.RandomState(0)
pairs = [((1080, 1920), (480, 640)), ((480, 640), (480, 640)), ((7, 9), (3, 4)), ((1, 1), (1, 1)), ((1, 1), (5, 7)), ((2, 3), (40, 64)),
         ((100, 1), (10, 1)), ((1, 100), (1, 7)), ((480, 640), (1, 1)), ((1000, 1000), (3, 2)), ((33, 47), (32, 46)), ((33, 47), (34, 48)),
         ((5, 5), (0, 5)), ((5, 5), (-1, 3)), ((64, 64), (1080, 1920))]
for (sh, sw), (dh, dw) in pairs:
    img = rng.randint(0, 256, (sh, sw, 3)).astype(np.uint8)
    for mode in ("linear", "area"):
        tag = "%dx%d -> %dx%d %s" % (sw, sh, dw, dh, mode)
        try:
            got = ctx.resize_bgr_to_gray(img, dw, dh, interpolation=mode).cpu().numpy()
            ref = orc.resize_bgr_to_gray(img, dw, dh) if mode == "linear" else orc.resize_area_bgr_to_gray(img, dw, dh)
            print("%-34s ok   identical %s (max diff %d)" % (tag, bool(np.array_equal(got, ref)), int(np.abs(got.astype(int) - ref.astype(int)).max()) if got.size else 0), flush=True)
        except RcflowError as e:
            print("%-34s error %d" % (tag, e.code), flush=True)
        except Exception as e:
            print("%-34s %s: %s" % (tag, type(e).__name__, str(e)[:70]), flush=True)
