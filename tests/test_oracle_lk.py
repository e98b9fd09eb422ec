"""CPU tests of the sparse PyrLK oracle (oracle/lk_oracle.cpp).

PARITY UNPINNED: the reference holds no vectors for cv::calcOpticalFlowPyrLK and OpenCV is absent;
the oracle is pinned by independent numpy restatements of its integer stages and by analytic
known-answer tests (recovery of a known translation).
"""
import numpy as np
from scipy.ndimage import correlate1d

from ripcurrents_amd import synth


def test_pyrdown_matches_numpy(orc):
    rng = np.random.RandomState(3)
    for (h, w) in ((240, 320), (135, 241), (7, 5)):
        img = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
        k = np.array([1, 4, 6, 4, 1])
        t = correlate1d(correlate1d(img.astype(np.int64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
        ref = ((t + 128) >> 8)[::2, ::2].astype(np.uint8)
        assert np.array_equal(orc.pyrdown_u8(img), ref)


def test_scharr_matches_numpy(orc):
    rng = np.random.RandomState(4)
    img = rng.randint(0, 256, size=(97, 131)).astype(np.uint8)
    d = orc.scharr_deriv(img).astype(np.int64)
    i64 = img.astype(np.int64)
    dx = correlate1d(correlate1d(i64, np.array([3, 10, 3]), axis=0, mode="mirror"), np.array([-1, 0, 1]), axis=1,
                     mode="mirror")
    dy = correlate1d(correlate1d(i64, np.array([-1, 0, 1]), axis=0, mode="mirror"), np.array([3, 10, 3]), axis=1,
                     mode="mirror")
    assert np.array_equal(d[..., 0], dx) and np.array_equal(d[..., 1], dy)


def test_level_rule(orc):
    # buildOpticalFlowPyramid stops when the NEXT level would not be larger than the window
    assert orc.pyrlk_levels(1920, 1080, (50, 50), 3) == 3
    assert orc.pyrlk_levels(320, 240, (50, 50), 3) == 2       # 40x30 would be <= 50
    assert orc.pyrlk_levels(320, 240, (21, 21), 3) == 3
    assert orc.pyrlk_levels(64, 64, (50, 50), 3) == 0


def test_translation_recovered(orc):
    fr = synth.translating_clip(320, 240, 2)          # constant (1.25, -0.75) px/frame
    pts = np.array([[160, 120], [100, 80], [250, 200], [30, 30], [290, 40]], np.float32)
    for win, eps in (((21, 21), 0.01), ((50, 50), 0.1)):
        q, st, er = orc.pyrlk(fr[0], fr[1], pts, win=win, max_level=3, epsilon=eps, flags=0)
        assert st.all()
        d = q - pts
        assert np.abs(d[:, 0] - 1.25).max() < 0.03 and np.abs(d[:, 1] + 0.75).max() < 0.03
        assert (er > 0).all() and (er < 2.0).all()     # mean absolute residual in grey levels


def test_status_and_min_eig(orc):
    fr = synth.translating_clip(320, 240, 2)
    flat = np.full((240, 320), 77, np.uint8)
    pts = np.array([[160, 120], [-200, 50], [100, 5000]], np.float32)
    q, st, er = orc.pyrlk(fr[0], fr[1], pts, win=(21, 21), flags=8)
    assert list(st) == [1, 0, 0]
    assert er[0] > 1e-4                                 # GET_MIN_EIGENVALS: err = min eigenvalue
    # textureless image: min eigenvalue 0 < threshold -> status 0, point unchanged
    q, st, er = orc.pyrlk(flat, flat, pts[:1], win=(21, 21), flags=8)
    assert st[0] == 0 and er[0] == 0 and np.array_equal(q[0], pts[0])
    # OPTFLOW_USE_INITIAL_FLOW starts from the supplied guess
    guess = pts[:1] + np.array([[1.0, -1.0]], np.float32)
    q2, st2, _ = orc.pyrlk(fr[0], fr[1], pts[:1], next_pts=guess, win=(21, 21), flags=4)
    assert st2[0] == 1 and np.abs(q2[0] - pts[0] - (1.25, -0.75)).max() < 0.03


def test_streakline_step_lk(orc):
    fr = synth.translating_clip(320, 240, 4)
    verts = np.zeros((16, 2), np.float32)
    verts[0] = (150.0, 100.0)
    n, fc = 1, 1
    for t in range(3):
        n, fc = orc.streakline_step_lk(verts, n, (150.0, 100.0), fr[t], fr[t + 1], fc)
    assert n == 4 and fc == 4
    assert tuple(verts[0]) == (150.0, 100.0)
    # the oldest vertex has been carried by three frames of (1.25, -0.75)
    assert np.abs(verts[3] - (150 + 3 * 1.25, 100 - 3 * 0.75)).max() < 0.1
