#!/usr/bin/env python3
"""Secondary measurements (not the headline): per-kernel throughput of the analysis rows
B1-B5 at 1080p and 4K, BASELINE config 3 (3840x2160, 5 scales + advection), the Gaussian
window variants of main.cpp, and the PCIe-inclusive host-pointer call."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters


def main():
    out = {}
    dev = torch.device("cuda")
    for (W, H, levels, tag) in ((1920, 1080, 2, "1080p"), (3840, 2160, 4, "4k")):
        ctx = Context(W, H)
        P = dict(pyr_scale=0.5, levels=levels, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
        T = 17 if tag == "1080p" else 9
        frames = synth.surf_clip(W, H, T, device=dev)
        flows = torch.empty((T - 1, H, W, 2), dtype=torch.float32, device=dev)
        n = W * H
        ctx.analysis_reset(W, H)
        r = {}
        t = timeit(lambda: ctx.farneback_clip(frames, flows, **P), 5, 2) / (T - 1)
        r["farneback_%dscales_us_per_frame" % (levels + 1)] = round(t * 1e6, 1)
        r["farneback_fps"] = round(1 / t, 1)
        f0 = flows[0]
        t = timeit(lambda: ctx.histogram_accumulate(f0)); r["histogram_GBs"] = round(8 * n / t / 1e9, 1)
        ctx.thresholds()
        t = timeit(lambda: ctx.create_flow_accumulate(f0, 40, want=("outmask",)))
        r["classify_accumulate_GBs"] = round(17 * n / t / 1e9, 1)
        t = timeit(lambda: ctx.streamline_field(f0, 2.0, 1))
        r["streamline_field_GBs"] = round(32 * n / t / 1e9, 1)
        pts = torch.rand((250, 2), device=dev) * torch.tensor([W - 4.0, H - 4.0], device=dev) + 2
        t = timeit(lambda: ctx.streamline(pts.clone(), f0, 2.0, 1, 100.0, variant=3))
        r["streamline_250_seeds_us"] = round(t * 1e6, 1)
        # the whole per-frame analysis of ripcurrents.cpp:229-439 on a resident flow field
        def frame_analysis():
            ctx.streamline_field(f0, 2.0, 1)
            ctx.histogram_accumulate(f0)
            ctx.thresholds()
            ctx.create_flow_accumulate(f0, 40, want=("outmask",))
        t = timeit(frame_analysis)
        r["analysis_us_per_frame"] = round(t * 1e6, 1)
        r["analysis_GBs_of_57B_per_px"] = round(57 * n / t / 1e9, 1)
        if tag == "1080p":
            for name, p in (("main264_gauss_win3", dict(P, flags=256)), ("main1119_gauss_win10_it3", dict(P, winsize=10, iterations=3, flags=256)),
                            ("main609_gauss_win20_it3", dict(P, winsize=20, iterations=3, flags=256)),
                            ("android_box_win5_it3_4scales", dict(P, levels=3, winsize=5, iterations=3, flags=0))):
                t = timeit(lambda: ctx.farneback_clip(frames, flows, **p), 3, 1) / (T - 1)
                r[name + "_fps"] = round(1 / t, 1)
            host = frames[:2].cpu().numpy()
            hf = np.empty((H, W, 2), np.float32)
            t0 = time.perf_counter()
            for _ in range(10):
                ctx.calcOpticalFlowFarneback(host[0], host[1], hf, **P)
            r["host_pointer_call_fps_pcie_inclusive"] = round(10 / (time.perf_counter() - t0), 1)
        out[tag] = r
        ctx.close()
        del frames, flows
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
