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
    flow = oracle.farneback(clip[0], clip[1], **p)
    np.savez_compressed(os.path.join(HERE, "farneback_rc215_96x80.npz"), prev=clip[0], next=clip[1], flow=flow, **p)
    p2 = dict(p, flags=256, winsize=10, iters=3)
    flow2 = oracle.farneback(clip[0], clip[1], **p2)
    np.savez_compressed(os.path.join(HERE, "farneback_main1119_96x80.npz"), prev=clip[0], next=clip[1], flow=flow2, **p2)
    st = oracle.HistState()
    oracle.create_histogram(oracle.flow_to_polar(flow), st)
    np.savez_compressed(os.path.join(HERE, "histogram_96x80.npz"), flow=flow, hist=st.hist, hist2d=st.hist2d,
                        histsum=np.int32(st.histsum.value), histsum2d=st.histsum2d, UPPER=np.float32(st.UPPER),
                        UPPER2d=st.UPPER2d, prop_above_upper=st.prop_above_upper)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
