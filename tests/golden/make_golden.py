#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/.

The reference (borgor/ripcurrents) ships no golden vectors for this path and its Farneback
arithmetic lives in un-vendored OpenCV, so these fixtures are NOT
reference outputs: they are seeded synthetic inputs and the outputs of this repository's CPU
oracle (oracle/liboracle.so), committed so that the oracle and the HIP path are pinned
against drift.  PARITY UNPINNED with respect to the reference.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle          # noqa: E402
from ripcurrents_amd import synth  # noqa: E402


def main():
    clip = synth.surf_clip(96, 80, 2, seed=1234)
    p = dict(pyr_scale=0.5, levels=2, winsize=3, iters=2, poly_n=15, poly_sigma=1.2, flags=0)
    # every Farneback fixture carries the conditioning of its pixels (determinant of the final solve, minimum along the
    # coarse-to-fine path), so that the GPU test applies SURVEY 8(d)'s conditioned metric to it
    flow, dl, dm = oracle.farneback_diag(clip[0], clip[1], **p)
    np.savez_compressed(os.path.join(HERE, "farneback_rc215_96x80.npz"), prev=clip[0], next=clip[1], flow=flow,
                        det_last=dl.astype(np.float32), det_min=dm, **p)
    p2 = dict(p, flags=256, winsize=10, iters=3)
    flow2, dl2, dm2 = oracle.farneback_diag(clip[0], clip[1], **p2)
    np.savez_compressed(os.path.join(HERE, "farneback_main1119_96x80.npz"), prev=clip[0], next=clip[1], flow=flow2,
                        det_last=dl2.astype(np.float32), det_min=dm2, **p2)
    # half the reference's working size (ripcurrents.hpp:4-5) with the parameters of ripcurrents.cpp:215: three full scales
    # (no level cropped), enough interior for the conditioned metric to bite, ~1 MB
    clip320 = synth.surf_clip(320, 240, 2, seed=4321)
    flow4, dl4, dm4 = oracle.farneback_diag(clip320[0], clip320[1], **p)
    np.savez_compressed(os.path.join(HERE, "farneback_rc215_320x240.npz"), prev=clip320[0], next=clip320[1], flow=flow4,
                        det_last=dl4.astype(np.float32), det_min=dm4, **p)
    # main.cpp:264 (Gaussian winsize 3): the library's default path for it is upstream's operation order, so this one
    # must be reproduced bit for bit by the HIP path; the determinants of the final solves travel with it
    p3 = dict(p, flags=256)
    flow3, det_last, det_min = oracle.farneback_diag(clip[0], clip[1], **p3)
    np.savez_compressed(os.path.join(HERE, "farneback_main264_96x80.npz"), prev=clip[0], next=clip[1], flow=flow3,
                        det_last=det_last.astype(np.float32), det_min=det_min, **p3)
    st = oracle.HistState()
    oracle.create_histogram(oracle.flow_to_polar(flow), st)
    np.savez_compressed(os.path.join(HERE, "histogram_96x80.npz"), flow=flow, hist=st.hist, hist2d=st.hist2d,
                        histsum=np.int32(st.histsum.value), histsum2d=st.histsum2d, UPPER=np.float32(st.UPPER),
                        UPPER2d=st.UPPER2d, prop_above_upper=st.prop_above_upper)
    # sparse PyrLK (Streakline.cpp:32 parameters) and the display path
    lk = synth.surf_clip(160, 120, 2, seed=77)
    rng = np.random.RandomState(5)
    pts = np.stack([rng.uniform(4, 156, 24), rng.uniform(4, 116, 24)], axis=1).astype(np.float32)
    pts[0] = (-50.0, 10.0)
    q, st2, er = oracle.pyrlk(lk[0], lk[1], pts, win=(50, 50), max_level=3, epsilon=0.1, flags=10)
    q21, st21, er21 = oracle.pyrlk(lk[0], lk[1], pts, win=(21, 21), max_level=3, epsilon=0.01, flags=0)
    np.savez_compressed(os.path.join(HERE, "pyrlk_160x120.npz"), prev=lk[0], next=lk[1], pts=pts, next50=q, status50=st2,
                        err50=er, next21=q21, status21=st21, err21=er21)
    pt = (rng.randn(48, 64, 2) * 2).astype(np.float32)
    dist = (np.abs(rng.randn(48, 64)) * 3).astype(np.float32)
    dist[0, :5] = 0
    pt[0, :3] = 0
    imgs = [oracle.streamline_display(pt, dist, w)[0] for w in (0, 1, 2)]
    hsv = np.stack([rng.uniform(0, 360, (48, 64)), rng.choice([0.7, 1.0], (48, 64)), rng.uniform(0, 1.5, (48, 64))],
                   axis=2).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "display_64x48.npz"), pt=pt, dist=dist, displacement=imgs[0], total_motion=imgs[1],
                        ratio=imgs[2], positions=oracle.streamline_positions(pt), jet=oracle.jet_lut(), hsv=hsv,
                        bgr=oracle.hsv_to_bgr(hsv))
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
