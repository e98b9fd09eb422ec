"""GPU parity tests for the per-pixel analysis downstream of the flow field (B1-B8):
librcflow (HIP, through the C ABI) vs the CPU oracle on identical flow input.

Bar: bit-exact for counts, thresholds, classes, masks, accumulators and particle
positions (integer work, or float work evaluated operation by operation in the reference's
order).  Reductions that the reference does as a sequential float/double running sum
(subtructMeanMagnitude, cv::mean) are compared with a stated tolerance.
"""
import numpy as np
import pytest
import torch

from ripcurrents_amd import RcflowError, synth
from ripcurrents_amd.api import HistState, Streakline

pytestmark = pytest.mark.gpu


def _flow_field(w, h, seed, scale=1.0):
    """A smooth field with the value range of real flows plus a few hazards: exact zeros,
    magnitudes beyond the last bin, negative-zero and axis-aligned vectors."""
    rng = np.random.RandomState(seed)
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    f = np.zeros((h, w, 2), np.float32)
    f[..., 0] = scale * (0.8 * np.sin(xs / 37.0) + 0.3 * np.cos(ys / 11.0) + 0.05 * rng.randn(h, w))
    f[..., 1] = scale * (0.6 * np.cos(xs / 23.0 + ys / 31.0) + 0.05 * rng.randn(h, w))
    f[5:9, 5:40] = 0.0
    f[10, 3:20, 0] = 3.0; f[10, 3:20, 1] = -1e-9        # angle rounds to 360.0f -> direction 36 -> 0
    f[12, 3:20, 0] = -0.0; f[12, 3:20, 1] = 0.7
    f[14, 3:20] = (9.0, 9.0)                             # beyond bin 49: not counted
    f[16, 3:20] = (0.0, -1.3)
    return f


def test_fast_atan_and_histogram_exact(ctx, orc):
    w, h = 640, 480
    st, ost = HistState(), orc.HistState()
    ctx.analysis_reset(w, h)
    for t in range(3):    # cumulative over frames, never reset (ripcurrents.cpp:147-154)
        f = _flow_field(w, h, t, scale=1.0 + 0.5 * t)
        ctx.create_histogram(f, st)
        orc.create_histogram(orc.flow_to_polar(f), ost)
        assert np.array_equal(st.hist, ost.hist)
        assert np.array_equal(st.hist2d, ost.hist2d)
        assert st.histsum == ost.histsum.value
        assert np.array_equal(st.histsum2d, ost.histsum2d)
        assert st.UPPER == ost.UPPER
        assert np.array_equal(st.UPPER2d, ost.UPPER2d)
        assert np.array_equal(st.prop_above_upper, ost.prop_above_upper, equal_nan=True)
    assert st.histsum > 0.9 * 3 * w * h


def test_histogram_edge_cases(ctx, orc):
    # empty histogram: every magnitude beyond the last bin -> prop is 0/0 = NaN as in the reference
    w, h = 96, 64
    f = np.full((h, w, 2), 50.0, np.float32)
    st, ost = HistState(), orc.HistState()
    ctx.analysis_reset(w, h)
    ctx.create_histogram(f, st)
    orc.create_histogram(orc.flow_to_polar(f), ost)
    assert st.histsum == 0 and ost.histsum.value == 0
    assert st.UPPER == ost.UPPER
    assert np.array_equal(st.prop_above_upper, ost.prop_above_upper, equal_nan=True)
    assert np.isnan(st.prop_above_upper).all()
    # ragged width, strided rows, all in one bin
    w, h = 333, 17
    big = np.zeros((h, 400, 2), np.float32)
    big[:, :w] = (0.31, 0.0)
    st, ost = HistState(), orc.HistState()
    ctx.analysis_reset(w, h)
    ctx.create_histogram(torch.as_tensor(big).cuda()[:, :w], st)
    orc.create_histogram(orc.flow_to_polar(np.ascontiguousarray(big[:, :w])), ost)
    assert np.array_equal(st.hist2d, ost.hist2d) and st.hist[6] == w * h


def test_histogram_key_on_bin_edges(ctx, orc):
    """Adversarial field for the bin / direction floors: magnitudes at k/20 and directions at multiples of
    10 degrees to within a few ulps, axis-aligned and diagonal vectors, zeros, denormals, huge values, Inf
    and NaN (a NaN magnitude is not counted: it converts to INT_MIN on the reference's x86).  Counts must
    equal the oracle's exactly."""
    w, h = 2048, 1024
    n = w * h
    rng = np.random.RandomState(11)
    f = np.empty((n, 2), np.float32)
    q = n // 8
    # 1: magnitudes on the bin edges, random direction
    k = rng.randint(1, 51, q).astype(np.float64) / 20.0
    th = rng.rand(q) * 2 * np.pi
    f[:q, 0] = k * np.cos(th); f[:q, 1] = k * np.sin(th)
    # 2: directions on the 10-degree edges, random magnitude, then nudged by a few ulps
    m = rng.randint(0, 37, q) * (np.pi / 18)
    r = rng.rand(q) * 2.6
    f[q:2 * q, 0] = r * np.cos(m); f[q:2 * q, 1] = r * np.sin(m)
    nudged = f[q:2 * q].view(np.int32) + rng.randint(-3, 4, (q, 2)).astype(np.int32)
    f[q:2 * q] = nudged.view(np.float32)
    # 3: both at once
    k = rng.randint(1, 51, q).astype(np.float64) / 20.0
    m = rng.randint(0, 37, q) * (np.pi / 18)
    f[2 * q:3 * q, 0] = k * np.cos(m); f[2 * q:3 * q, 1] = k * np.sin(m)
    # 4: axis-aligned / diagonal with magnitudes on a fine lattice
    v = (rng.randint(-60, 61, (q, 2)) * 0.05).astype(np.float32)
    v[rng.rand(q) < 0.3, 0] = 0
    v[rng.rand(q) < 0.3, 1] = 0
    f[3 * q:4 * q] = v
    # 5: ordinary smooth and random vectors
    f[4 * q:6 * q] = rng.randn(2 * q, 2) * 0.8
    f[6 * q:7 * q] = (rng.rand(q, 2) - 0.5) * 5.2
    # 6: extremes
    ex = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39, 1e-30, 1e-20, 1e-10, 2.45, 2.5, 2.55, 1e10, 1e20, 3e38, np.inf, -np.inf,
                   np.nan, 0.05, 0.1, 1.0], np.float32)
    f[7 * q:, 0] = ex[rng.randint(0, len(ex), n - 7 * q)]
    f[7 * q:, 1] = ex[rng.randint(0, len(ex), n - 7 * q)]
    f = f[rng.permutation(n)].reshape(h, w, 2)
    st, ost = HistState(), orc.HistState()
    ctx.analysis_reset(w, h)
    with np.errstate(all="ignore"):
        ctx.create_histogram(f, st)
        orc.create_histogram(orc.flow_to_polar(f), ost)
    assert st.histsum == ost.histsum.value and st.histsum > n // 2
    assert np.array_equal(st.hist2d, ost.hist2d)
    assert st.UPPER == ost.UPPER and np.array_equal(st.UPPER2d, ost.UPPER2d)


def _adversarial_flow(w, h, seed):
    """Ordinary vectors mixed with zeros, denormals, magnitudes exactly on the 0.2 / 0.5 class thresholds,
    huge values, Inf and NaN."""
    n = w * h
    rng = np.random.RandomState(seed)
    ex = np.array([0.0, -0.0, 1e-45, 1e-39, 1e-20, 0.2, 0.5, 0.05, 1.0, 1.7, 2.5, 1e10, 3e38, np.inf, -np.inf, np.nan],
                  np.float32)
    f = (rng.randn(n, 2) * 0.8).astype(np.float32)
    for c in (0, 1):
        idx = rng.rand(n) < 0.3
        f[idx, c] = ex[rng.randint(0, len(ex), idx.sum())]
    k = rng.rand(n) < 0.1
    th = rng.rand(k.sum()) * 2 * np.pi
    r = np.where(rng.rand(k.sum()) < 0.5, 0.2, 0.5)
    f[k, 0] = (r * np.cos(th)).astype(np.float32)
    f[k, 1] = (r * np.sin(th)).astype(np.float32)
    return f.reshape(h, w, 2)


def test_analysis_on_adversarial_flow(ctx, orc):
    """Every per-pixel analysis kernel on a field with huge, infinite and NaN vectors: same results as the
    oracle, whose `(int)float` conversions are the reference's x86 ones (NaN / out of range -> INT_MIN).  A
    particle carried to 3e38 must be rejected by the sampler's bounds check like there, not wrap it."""
    w, h = 512, 256
    f = _adversarial_flow(w, h, 5)
    eq = lambda a, b: np.array_equal(a, b, equal_nan=True)
    with np.errstate(all="ignore"):
        ctx.analysis_reset(w, h)
        st, ost = HistState(), orc.HistState()
        ctx.create_histogram(f, st)
        polar = orc.flow_to_polar(f)
        orc.create_histogram(polar, ost)
        assert eq(st.hist2d, ost.hist2d) and st.UPPER == ost.UPPER and eq(st.UPPER2d, ost.UPPER2d)
        acc = np.zeros((h, w, 3), np.float32)
        for fc in (1, 31, 32, 40):
            outs = ctx.create_flow_accumulate(f, fc)
            wc = np.zeros((h, w, 3), np.float32)
            acc2 = np.zeros((h, w, 3), np.float32)
            p2 = polar.copy()
            orc.create_flow(p2, wc, acc2, ost.UPPER, 0.5, 0.2, ost.UPPER2d)
            out = np.zeros((h, w, 3), np.float32)
            mask = np.zeros((h, w), np.uint8)
            orc.create_accumulationbuffer(acc, acc2, out, mask, fc)
            assert eq(outs["waterclass"].cpu().numpy(), wc) and eq(outs["polar"].cpu().numpy(), p2)
            assert eq(outs["out"].cpu().numpy(), out) and eq(outs["outmask"].cpu().numpy(), mask)
            assert eq(ctx.accumulator(w, h), acc[..., 0])
        ctx.analysis_reset(w, h)
        pt = np.zeros((h, w, 2), np.float32)
        dist = np.zeros((h, w), np.float32)
        for _ in range(3):
            ctx.streamline_field(f, 2.0, 2, UPPER=1.7)
            orc.streamline_field(pt, dist, f, 2.0, 2, 1.7)
        gpt, gdist = ctx.streamline_field_state(w, h)
        assert eq(gpt, pt) and eq(gdist, dist)
        pts = (np.random.RandomState(6).rand(300, 2) * [w, h]).astype(np.float32)
        for variant in range(5):          # variant 2 has no cutoff: positions run off to huge values
            g, _ = ctx.streamline(pts.copy(), f, 0.1, 20, 1.7, variant=variant)
            o = pts.copy()
            orc.streamline_points(o, f, 0.1, 20, 1.7, variant=variant)
            assert eq(g.cpu().numpy() if hasattr(g, "cpu") else np.asarray(g), o)
        # colour / display rows on the same field
        pt0 = np.zeros((h, w, 2), np.float32)
        ref = pt0.copy()
        for _ in range(3):
            orc.get_delta_field(ref, f, 2.0, 1.8)
        d = torch.as_tensor(pt0).cuda()
        for _ in range(3):
            d = ctx.get_delta_field(d, f, 2.0, 1.8)
        assert eq(d.cpu().numpy(), ref)
        r, mf = orc.shear_rate_to_color(f, 0.0)
        g, gmf = ctx.shearRateToColor(f, 0.0)
        assert eq(np.float32(gmf), np.float32(mf)) and eq(g.cpu().numpy(), r)
        # hues far outside [0, 360), infinite and NaN: upstream's wrap loop would never end on the infinite
        # ones; kernel and oracle take at most 64 steps and then hue 0
        hsv = np.stack([f[..., 0] * 100, np.abs(f[..., 1]), f[..., 0]], -1).astype(np.float32)
        assert eq(ctx.hsv_to_bgr(hsv).cpu().numpy(), orc.hsv_to_bgr(hsv))
        ctx.analysis_reset(w, h)
        for _ in range(3):
            ctx.streamline_field(f, 2.0, 2, UPPER=1e30)
        spt, sdist = ctx.streamline_field_state(w, h)
        assert not np.isfinite(spt).all()
        for which in (0, 1, 2):
            img, mx = ctx.streamline_display(which)
            r, rmx = orc.streamline_display(spt, sdist, which)
            assert eq(np.float32(mx), np.float32(rmx)) and eq(img.cpu().numpy(), r)
        assert eq(ctx.streamline_positions().cpu().numpy(), orc.streamline_positions(spt))


def test_advection_iteration_count_is_bounded(ctx):
    """The step loops run on the device: an absurd count is an argument error, not a kernel that never returns."""
    from ripcurrents_amd import RcflowError
    w, h = 64, 48
    f = np.zeros((h, w, 2), np.float32)
    ctx.analysis_reset(w, h)
    for bad in (-1, 65537, 2 ** 31 - 1):
        with pytest.raises(RcflowError) as e:
            ctx.streamline_field(f, 2.0, bad, UPPER=1.0)
        assert e.value.code == -1
        with pytest.raises(RcflowError) as e:
            ctx.streamline(np.zeros((3, 2), np.float32), f, 0.1, bad, 1.0)
        assert e.value.code == -1
    ctx.streamline_field(f, 2.0, 65536, UPPER=1.0)      # zero field: every particle stops at once
    ctx.sync()


def test_histogram_random_directions(ctx, orc):
    """Uniformly random vectors: every wave holds ~64 distinct bins (the slow path of the
    ballot grouping)."""
    w, h = 512, 256
    rng = np.random.RandomState(3)
    f = (rng.rand(h, w, 2).astype(np.float32) - 0.5) * 5
    st, ost = HistState(), orc.HistState()
    ctx.analysis_reset(w, h)
    ctx.create_histogram(f, st)
    orc.create_histogram(orc.flow_to_polar(f), ost)
    assert np.array_equal(st.hist2d, ost.hist2d) and np.array_equal(st.UPPER2d, ost.UPPER2d)


@pytest.mark.parametrize("form", ["plain", "odd_width", "ragged_height", "unaligned_base", "unaligned_pitch"])
def test_histogram_plain_and_general_forms(ctx, orc, form):
    """The histogram kernel has a plain form (even width, height a multiple of its four-row items, 16-byte aligned rows: one
    offset per item, loads issued unconditionally) and a general one; both count what the oracle counts, hazards included
    (NaN / infinite vectors, exact zeros, magnitudes beyond the last bin, vectors on bin and direction edges)."""
    w, h = {"plain": (640, 480), "odd_width": (639, 480), "ragged_height": (640, 478),
            "unaligned_base": (638, 480), "unaligned_pitch": (640, 480)}[form]
    rng = np.random.RandomState(11)
    f = _flow_field(w, h, 7, scale=1.7)
    f[20, 5:60] = (np.nan, 0.3)
    f[21, 5:60] = (np.inf, -np.inf)
    f[22, 5:60] = (-np.inf, 0.0)
    f[23, 5:60, 0] = np.arange(55, dtype=np.float32) * 0.05         # magnitudes on the bin edges, direction 0 exactly
    f[23, 5:60, 1] = 0.0
    ang = np.deg2rad(np.arange(55, dtype=np.float32) * 10.0)         # directions on the 10-degree edges
    f[24, 5:60, 0] = 1.234 * np.cos(ang); f[24, 5:60, 1] = 1.234 * np.sin(ang)
    f[h - 1, :] = (rng.rand(w, 2).astype(np.float32) - 0.5) * 6      # the last row (of a partial item when the height is ragged)
    if form == "unaligned_base":
        big = torch.zeros((h, w + 2, 2), dtype=torch.float32, device="cuda")
        big[:, 1:w + 1] = torch.as_tensor(f).cuda()
        dev = big[:, 1:w + 1]                                         # rows start 8 bytes off a 16-byte boundary
        assert dev.data_ptr() % 16 == 8
    elif form == "unaligned_pitch":
        big = torch.zeros((h, w + 1, 2), dtype=torch.float32, device="cuda")
        big[:, :w] = torch.as_tensor(f).cuda()
        dev = big[:, :w]                                              # pitch 8 (w + 1) bytes: every other row unaligned
        assert (dev.stride(0) * 4) % 16 == 8
    else:
        dev = torch.as_tensor(f).cuda()
    st, ost = HistState(), orc.HistState()
    ctx.analysis_reset(w, h)
    ctx.create_histogram(dev, st)
    orc.create_histogram(orc.flow_to_polar(f), ost)
    assert np.array_equal(st.hist, ost.hist)
    assert np.array_equal(st.hist2d, ost.hist2d)
    assert st.histsum == ost.histsum.value
    assert np.array_equal(st.histsum2d, ost.histsum2d)
    assert st.UPPER == ost.UPPER and np.array_equal(st.UPPER2d, ost.UPPER2d)


def test_classify_accumulate_exact(ctx, orc):
    w, h = 320, 240
    ctx.analysis_reset(w, h)
    ost = orc.HistState()
    acc = np.zeros((h, w, 3), np.float32)
    for framecount in (1, 2, 31, 32, 33, 40):
        f = _flow_field(w, h, framecount, scale=1.5)
        polar = orc.flow_to_polar(f)
        st = HistState()
        ctx.create_histogram(f, st)
        orc.create_histogram(polar, ost)
        outs = ctx.create_flow_accumulate(f, framecount)
        wc = np.zeros((h, w, 3), np.float32)
        acc2 = np.zeros((h, w, 3), np.float32)
        orc.create_flow(polar, wc, acc2, ost.UPPER, 0.5, 0.2, ost.UPPER2d)
        out = np.zeros((h, w, 3), np.float32)
        mask = np.zeros((h, w), np.uint8)
        orc.create_accumulationbuffer(acc, acc2, out, mask, framecount)
        assert np.array_equal(outs["waterclass"].cpu().numpy(), wc)
        assert np.array_equal(outs["polar"].cpu().numpy(), polar)       # polar rewritten for display
        assert np.array_equal(outs["out"].cpu().numpy(), out)
        assert np.array_equal(outs["outmask"].cpu().numpy(), mask)
        assert np.array_equal(ctx.accumulator(w, h), acc[..., 0])
    assert acc[..., 0].max() >= 1.0


def test_streamline_field_exact(ctx, orc):
    w, h = 320, 240
    ctx.analysis_reset(w, h)
    pt = np.zeros((h, w, 2), np.float32)
    dist = np.zeros((h, w), np.float32)
    for t in range(4):
        f = _flow_field(w, h, 10 + t, scale=2.0)
        ctx.streamline_field(f, 2.0, 1, UPPER=1.7)
        orc.streamline_field(pt, dist, f, 2.0, 1, 1.7)
    gpt, gdist = ctx.streamline_field_state(w, h)
    assert np.array_equal(gpt, pt) and np.array_equal(gdist, dist)
    assert np.abs(pt).max() > 3.0
    # several iterations per call, slot UPPER (the value the histogram produced)
    ctx.analysis_reset(w, h)
    st = HistState()
    f = _flow_field(w, h, 99, scale=2.0)
    ctx.create_histogram(f, st)
    ctx.streamline_field(f, 1.5, 3)            # UPPER < 0 -> slot's UPPER
    pt = np.zeros((h, w, 2), np.float32)
    dist = np.zeros((h, w), np.float32)
    orc.streamline_field(pt, dist, f, 1.5, 3, st.UPPER)
    gpt, gdist = ctx.streamline_field_state(w, h)
    assert np.array_equal(gpt, pt) and np.array_equal(gdist, dist)


@pytest.mark.parametrize("variant,dt,iters", [(0, 0.1, 100), (1, 0.5, 30), (2, 0.1, 100), (3, 2.0, 1), (4, 2.0, 4)])
def test_streamline_points_exact(ctx, orc, variant, dt, iters):
    w, h = 320, 240
    f = _flow_field(w, h, 5, scale=2.0)
    rng = np.random.RandomState(0)
    pts = np.stack([rng.randint(0, w, 250), rng.randint(0, h, 250)], 1).astype(np.float32)  # ripcurrents.cpp:174-176
    pts[0] = (0.5, 0.5)          # rejected by the bounds check
    pts[1] = (w - 1.5, 10.0)     # xind + 2 > cols
    ref = pts.copy()
    tr_ref = orc.streamline_points(ref, f, dt, iters, 1.9, variant, trace=True)
    got, tr = ctx.streamline(pts, f, dt, iters, 1.9, variant, trace=True)
    assert np.array_equal(got.cpu().numpy(), ref)
    assert np.array_equal(tr.cpu().numpy(), tr_ref)


def test_rotation_field_known_answer(ctx):
    """validate_streamlines (main.cpp:372-435): the analytic rotational field; forward Euler
    with dt = 0.03 conserves X^2/W + Y^2/H up to the factor 1 + (100 dt)^2/(W H) per step."""
    w, h = 640, 480
    f = synth.rotation_field(w, h)
    pts = np.array([[200.0, 200.0]], np.float32)
    steps = 700
    got, tr = ctx.streamline(pts, f, 0.03, steps, 1e9, variant=0, trace=True)
    tr = tr.cpu().numpy()[0].astype(np.float64)
    inv = (tr[:, 0] - w / 2) ** 2 / w + (tr[:, 1] - h / 2) ** 2 / h
    inv0 = (200 - w / 2) ** 2 / w + (200 - h / 2) ** 2 / h
    growth = (1 + (100 * 0.03) ** 2 / (w * h)) ** np.arange(1, steps + 1)
    assert np.abs(inv / (inv0 * growth) - 1).max() < 2e-3
    # it orbits: the angle around the centre advances monotonically
    ang = np.unwrap(np.arctan2((tr[:, 1] - h / 2) / np.sqrt(h), (tr[:, 0] - w / 2) / np.sqrt(w)))
    assert (np.diff(ang) > 0).all()


def test_get_delta_field_exact(ctx, orc):
    w, h = 200, 150
    f = _flow_field(w, h, 8, scale=2.0)
    pt = np.zeros((h, w, 2), np.float32)
    ref = pt.copy()
    for _ in range(3):
        orc.get_delta_field(ref, f, 2.0, 1.8)
    d = torch.as_tensor(pt).cuda()
    for _ in range(3):
        d = ctx.get_delta_field(d, f, 2.0, 1.8)
    assert np.array_equal(d.cpu().numpy(), ref)


def test_streakline_bookkeeping(ctx, orc):
    w, h = 320, 240
    f = _flow_field(w, h, 2, scale=3.0)
    f[100:104, 150:154] = 80.0          # a jump > 0.1*W that must be rejected (Streakline.cpp:35-40)
    s = Streakline((152.0, 102.0))
    verts = np.zeros((64, 2), np.float32)
    verts[0] = (152.0, 102.0)
    n, fc = 1, 1
    for _ in range(12):
        s.run(ctx, f, w, h, dt=1.0)
        n, fc = orc.streakline_step(verts, n, (152.0, 102.0), f, 1.0, fc)
        assert s.numberOfVertices == n and s.frameCount == fc
        assert np.array_equal(np.asarray(s.vertices, np.float32), verts[:n])
    assert s.vertices[0] == (152.0, 102.0) and n == 13


def test_flow_postops(ctx, orc):
    w, h = 640, 480
    f = _flow_field(w, h, 4, scale=1.2)
    # subtructAverage: cv::mean in double, subtraction in double -> at most 1 ulp apart
    ref = f.copy(); orc.subtract_average(ref)
    got = ctx.subtructAverage(f.copy()).cpu().numpy()
    assert np.abs(got - ref).max() <= 2.4e-7 * max(1.0, np.abs(ref).max())
    # stabilizer: small patch, fixed summation order
    ref = f.copy(); orc.stabilizer(ref)
    got = ctx.stabilizer(f.copy()).cpu().numpy()
    assert np.abs(got - ref).max() <= 2.4e-7 * max(1.0, np.abs(ref).max())
    # subtructMeanMagnitude: the reference's mean is a sequential float32 running sum over
    # 307200 pixels (relative error ~1e-5); the HIP path sums in double.  Stated tolerance:
    # 1e-4 relative on the mean, hence 1e-4 * mean absolute on the result.
    ref = f.copy(); orc.subtract_mean_magnitude(ref)
    got = ctx.subtructMeanMagnitude(f.copy()).cpu().numpy()
    mean_mag = np.sqrt((f.astype(np.float64) ** 2).sum(-1)).mean()
    assert np.abs(got - ref).max() <= 1e-4 * mean_mag + 1e-6
    # sliding-window mean (main.cpp:1142-1153): elementwise, exact
    avg = np.zeros((h, w, 2), np.float32); slot = np.zeros_like(avg)
    davg, dslot = torch.as_tensor(avg).cuda(), torch.as_tensor(slot).cuda()
    for t in range(3):
        cur = _flow_field(w, h, 20 + t)
        orc.window_mean_update(avg, slot, cur, 10)
        ctx.window_mean(davg, dslot, torch.as_tensor(cur).cuda(), 10)
    assert np.array_equal(davg.cpu().numpy(), avg) and np.array_equal(dslot.cpu().numpy(), slot)


def test_colouring(ctx, orc):
    w, h = 320, 240
    md = 0.0
    gmd = 0.0
    for t in range(2):     # first frame divides by max_displacement == 0 like the reference
        f = _flow_field(w, h, 30 + t, scale=2.0)
        ref, md = orc.vector_to_color(f, md)
        got, gmd = ctx.vectorToColor(f, gmd)
        got = got.cpu().numpy()
        assert gmd == md
        # hue byte included: the angle is the correctly rounded float atan2 on both sides (double atan2 rounded once)
        assert np.array_equal(got, ref)
    mf = gmf = 0.0
    for t in range(2):
        f = _flow_field(w, h, 40 + t, scale=2.0)
        ref, mf = orc.shear_rate_to_color(f, mf)
        got, gmf = ctx.shearRateToColor(f, gmf)
        assert gmf == mf
        assert np.array_equal(got.cpu().numpy(), ref)


def test_frame_loop_like_reference(ctx, orc):
    """The call order of ripcurrents.cpp:194-440 over a short clip: flow -> streamline_field
    (with the previous frame's UPPER) -> histogram -> classify/accumulate, GPU flows fed to
    both sides so the analysis must agree exactly frame by frame."""
    w, h, T = 320, 240, 5
    clip = torch.as_tensor(synth.surf_clip(w, h, T, seed=12)).cuda()
    ctx.analysis_reset(w, h)
    ctx.stream_reset()
    ost = orc.HistState()
    pt = np.zeros((h, w, 2), np.float32); dist = np.zeros((h, w), np.float32)
    acc = np.zeros((h, w, 3), np.float32)
    framecount = 0
    for t in range(T):
        flow = ctx.push_frame(clip[t])
        if flow is None:
            continue
        framecount += 1
        ctx.streamline_field(flow, 2.0, 1)                     # slot UPPER (100.0 on the first frame)
        st = HistState()
        ctx.create_histogram(flow, st)
        outs = ctx.create_flow_accumulate(flow, framecount, want=("outmask",))
        hf = flow.cpu().numpy()
        orc.streamline_field(pt, dist, hf, 2.0, 1, ost.UPPER)
        polar = orc.flow_to_polar(hf)
        orc.create_histogram(polar, ost)
        wc = np.zeros((h, w, 3), np.float32); acc2 = np.zeros((h, w, 3), np.float32)
        orc.create_flow(polar, wc, acc2, ost.UPPER, 0.5, 0.2, ost.UPPER2d)
        out = np.zeros((h, w, 3), np.float32); mask = np.zeros((h, w), np.uint8)
        orc.create_accumulationbuffer(acc, acc2, out, mask, framecount)
        assert st.UPPER == ost.UPPER and np.array_equal(st.hist2d, ost.hist2d)
        assert np.array_equal(outs["outmask"].cpu().numpy(), mask)
    gpt, gdist = ctx.streamline_field_state(w, h)
    assert np.array_equal(gpt, pt) and np.array_equal(gdist, dist)


def test_whole_frame_loop_on_device(ctx, orc):
    """ripcurrents.cpp:184-511 end to end with nothing but 8-bit images crossing the boundary: decoded
    BGR frame -> resize (INTER_AREA for the first frame, INTER_LINEAR after) + gray -> flow -> dense
    advection -> display images -> histogram/thresholds -> classify/accumulate -> edges -> output frame.
    The oracle follows with the GPU's flow field, so every later product must match exactly."""
    sw, sh, w, h, T = 960, 540, 320, 240, 4
    rng = np.random.RandomState(3)
    gray = synth.surf_clip(sw, sh, T, seed=15)
    bgr = np.stack([np.clip(gray.astype(np.int32) + d, 0, 255) for d in (-9, 0, 7)], axis=-1).astype(np.uint8)
    ctx.analysis_reset(w, h)
    ctx.stream_reset()
    ost = orc.HistState()
    pt = np.zeros((h, w, 2), np.float32); dist = np.zeros((h, w), np.float32)
    acc = np.zeros((h, w, 3), np.float32)
    framecount = 0
    for t in range(T):
        mode = "area" if t == 0 else "linear"                       # ripcurrents.cpp:186 vs :209
        f1 = ctx.resize_bgr_to_gray(bgr[t], w, h, interpolation=mode)
        ref_f1 = orc.resize_area_bgr_to_gray(bgr[t], w, h) if t == 0 else orc.resize_bgr_to_gray(bgr[t], w, h)
        assert np.array_equal(f1.cpu().numpy(), ref_f1)
        flow = ctx.push_frame(f1)                                    # :215 (+ u_f1.copyTo(u_f2))
        if flow is None:
            continue
        framecount += 1
        hf = flow.cpu().numpy()
        ctx.streamline_field(flow, 2.0, 1)                           # :229-231
        orc.streamline_field(pt, dist, hf, 2.0, 1, ost.UPPER)
        for which in (0, 1, 2):                                      # :233-257
            assert np.array_equal(ctx.streamline_display(which)[0].cpu().numpy(), orc.streamline_display(pt, dist, which)[0])
        st = HistState()
        ctx.create_histogram(flow, st)                               # :305-366
        polar = orc.flow_to_polar(hf)
        orc.create_histogram(polar, ost)
        outs = ctx.create_flow_accumulate(flow, framecount + 30, want=("outmask", "polar"))   # :376-439
        wc = np.zeros((h, w, 3), np.float32); acc2 = np.zeros((h, w, 3), np.float32)
        orc.create_flow(polar, wc, acc2, ost.UPPER, 0.5, 0.2, ost.UPPER2d)
        out = np.zeros((h, w, 3), np.float32); mask = np.zeros((h, w), np.uint8)
        orc.create_accumulationbuffer(acc, acc2, out, mask, framecount + 30)
        assert np.array_equal(ctx.hsv_to_bgr(outs["polar"]).cpu().numpy(), orc.hsv_to_bgr(polar))   # :405
        edges = ctx.create_edges(outs["outmask"])                     # :477-479
        ref_edges = orc.create_edges(mask)
        assert np.array_equal(edges.cpu().numpy(), ref_edges)
        subframe = rng.randint(0, 255, (h, w, 3)).astype(np.uint8)   # (the resized colour frame)
        got = ctx.create_output(subframe, edges).cpu().numpy()        # :487-505
        ref = subframe.copy(); ref[..., 2][ref_edges > 0] = 255
        assert np.array_equal(got, ref)
    assert framecount == T - 1


def test_create_edges_exact(ctx, orc):
    """5x5 ellipse dilate + morphological gradient on outmask (integer, exact)."""
    rng = np.random.RandomState(5)
    for (w, h) in ((320, 240), (333, 71), (64, 16), (7, 5)):
        m = (rng.rand(h, w) > 0.97).astype(np.uint8) * 255
        m[: h // 3, : w // 4] = 255                      # a blob touching the border
        m[h // 2, :] = rng.randint(0, 256, w)            # arbitrary grey values, not only 0/255
        got = ctx.create_edges(m).cpu().numpy()
        assert np.array_equal(got, orc.create_edges(m)), (w, h)
    assert np.array_equal(orc.ellipse5(), np.uint8([[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1],
                                                     [0, 0, 1, 0, 0]]))


def test_resize_bgr_to_gray_exact(ctx, orc):
    """resize INTER_LINEAR (fixed point) + BGR2GRAY on device (integer, exact)."""
    rng = np.random.RandomState(6)
    for (sw, sh, dw, dh) in ((1280, 720, 640, 480), (641, 479, 640, 480), (320, 240, 640, 480), (640, 480, 640, 480),
                             (97, 33, 31, 64)):
        f = rng.randint(0, 256, (sh, sw, 3)).astype(np.uint8)
        got = ctx.resize_bgr_to_gray(f, dw, dh).cpu().numpy()
        assert np.array_equal(got, orc.resize_bgr_to_gray(f, dw, dh)), (sw, sh, dw, dh)


def test_create_output_paints_the_edges_red(ctx):
    """create_output (ripcurrents_module.cpp:225-244): pixel.z = 255 where the edge mask is set, rest untouched."""
    rng = np.random.RandomState(2)
    for (w, h) in ((640, 480), (33, 7)):
        frame = rng.randint(0, 255, (h, w, 3)).astype(np.uint8)
        mask = (rng.rand(h, w) > 0.8).astype(np.uint8) * 255
        ref = frame.copy()
        ref[..., 2][mask > 0] = 255
        assert np.array_equal(ctx.create_output(frame, mask).cpu().numpy(), ref)


def test_resize_area_bgr_to_gray_exact(ctx, orc):
    """The first frame's INTER_AREA resize + BGR2GRAY (ripcurrents.cpp:186): integer factors (2 x 2, 3 x 3,
    3 x 2), the 1920x1080 -> 640x480 case (3 x 2.25) and ragged factors; bit-exact vs the oracle."""
    rng = np.random.RandomState(8)
    for (sw, sh, dw, dh) in ((1280, 960, 640, 480), (1920, 1440, 640, 480), (1920, 960, 640, 480), (1920, 1080, 640, 480),
                             (1000, 700, 640, 480), (641, 481, 640, 480), (640, 480, 640, 480), (97, 65, 31, 17)):
        f = rng.randint(0, 256, (sh, sw, 3)).astype(np.uint8)
        got = ctx.resize_bgr_to_gray(f, dw, dh, interpolation="area").cpu().numpy()
        assert np.array_equal(got, orc.resize_area_bgr_to_gray(f, dw, dh)), (sw, sh, dw, dh)
    with pytest.raises(Exception):
        ctx.resize_bgr_to_gray(np.zeros((240, 320, 3), np.uint8), 640, 480, interpolation="area")    # enlarging


def test_display_path_matches_oracle(ctx, orc):
    """streamline_displacement / _total_motion / _ratio / _positions and the float
    HSV->BGR of the flow window, on the slot's streamline field (ripcurrents.cpp:233-273, :405)."""
    w, h = 640, 480
    ctx.analysis_reset(w, h)
    for t in range(3):
        ctx.streamline_field(_flow_field(w, h, 20 + t, scale=1.5), 2.0, 1, UPPER=100.0)
    pt, dist = ctx.streamline_field_state(w, h)
    assert np.array_equal(ctx.jet_lut(), orc.jet_lut())
    for which in (0, 1, 2):
        img, mx = ctx.streamline_display(which)
        ref, rmx = orc.streamline_display(pt, dist, which)
        assert np.float32(rmx) == np.float32(mx)
        got = img.cpu().numpy()
        # sqrtf / division are correctly rounded on both sides: the 8-bit index is the same integer
        assert np.array_equal(got, ref), "which=%d: %d pixels differ" % (which, (got != ref).any(axis=2).sum())
    assert np.array_equal(ctx.streamline_positions().cpu().numpy(), orc.streamline_positions(pt))
    # the display image create_flow leaves in `current` (angle, 0.7|1, mag/UPPER2d) -> BGR
    rng = np.random.RandomState(5)
    hsv = np.stack([rng.uniform(0, 360, (h, w)), rng.choice([0.7, 1.0, 0.0], (h, w)), rng.uniform(0, 2, (h, w))],
                   axis=2).astype(np.float32)
    hsv[0, :8, 0] = (0, 60, 120, 180, 240, 300, 360, 359.99997)
    assert np.array_equal(ctx.hsv_to_bgr(hsv).cpu().numpy(), orc.hsv_to_bgr(hsv))


def test_global_thresholds_from_reduced_words(ctx, orc):
    """thresholds from the all-reduced counters of two segments equal the oracle's on
    the summed histogram, and the slot's own counters stay local."""
    import torch
    w, h = 640, 480
    words, sts = [], []
    for seg in range(2):
        ctx.analysis_reset(w, h)
        ost = orc.HistState()
        for t in range(2):
            f = _flow_field(w, h, 40 + 10 * seg + t, scale=1.0 + seg)
            ctx.histogram_accumulate(f)
            orc.histogram_accumulate(orc.flow_to_polar(f), ost)
        words.append(ctx.histogram_words().clone())
        sts.append(ost)
    total = words[0] + words[1]
    g = orc.HistState()
    g.hist = sts[0].hist + sts[1].hist; g.hist2d = sts[0].hist2d + sts[1].hist2d
    g.histsum.value = sts[0].histsum.value + sts[1].histsum.value; g.histsum2d = sts[0].histsum2d + sts[1].histsum2d
    orc.histogram_thresholds(g)
    ctx.thresholds_from_words(total)
    st = ctx.histogram_read()
    assert st.UPPER == g.UPPER and np.array_equal(st.UPPER2d, g.UPPER2d)
    assert np.array_equal(st.prop_above_upper, g.prop_above_upper, equal_nan=True)
    assert torch.equal(ctx.histogram_words(), words[1])          # local counters untouched


@pytest.mark.parametrize("size", [(1, 1), (2, 3), (33, 47), (130, 70), (65, 1), (1, 77)])
def test_analysis_rows_on_ragged_and_tiny_sizes(ctx, orc, size):
    """Every per-pixel analysis row on frames smaller than a block / with one row or column."""
    w, h = size
    rng = np.random.RandomState(w * 131 + h)
    f = (rng.randn(h, w, 2) * 0.8).astype(np.float32)
    ctx.analysis_reset(w, h)
    st, ost = HistState(), orc.HistState()
    ctx.create_histogram(f, st)
    polar = orc.flow_to_polar(f)
    orc.create_histogram(polar, ost)
    assert np.array_equal(st.hist2d, ost.hist2d) and st.histsum == ost.histsum.value
    assert st.UPPER == ost.UPPER and np.array_equal(st.prop_above_upper, ost.prop_above_upper, equal_nan=True)
    outs = ctx.create_flow_accumulate(f, 40)
    wc = np.zeros((h, w, 3), np.float32); acc2 = np.zeros((h, w, 3), np.float32)
    orc.create_flow(polar, wc, acc2, ost.UPPER, 0.5, 0.2, ost.UPPER2d)
    acc = np.zeros((h, w, 3), np.float32); out = np.zeros((h, w, 3), np.float32); mask = np.zeros((h, w), np.uint8)
    orc.create_accumulationbuffer(acc, acc2, out, mask, 40)
    assert np.array_equal(outs["waterclass"].cpu().numpy(), wc) and np.array_equal(outs["outmask"].cpu().numpy(), mask)
    assert np.array_equal(ctx.create_edges(mask).cpu().numpy(), orc.create_edges(mask))
    pt = np.zeros((h, w, 2), np.float32); dist = np.zeros((h, w), np.float32)
    for _ in range(2):
        ctx.streamline_field(f, 2.0, 1, UPPER=100.0)
        orc.streamline_field(pt, dist, f, 2.0, 1, 100.0)
    gpt, gdist = ctx.streamline_field_state(w, h)
    assert np.array_equal(gpt, pt) and np.array_equal(gdist, dist)
    for which in (0, 1, 2):
        img, mx = ctx.streamline_display(which)
        ref, rmx = orc.streamline_display(pt, dist, which)
        assert np.array_equal(img.cpu().numpy(), ref)
    assert np.array_equal(ctx.streamline_positions().cpu().numpy(), orc.streamline_positions(pt))
    seeds = np.stack([rng.uniform(-2, w + 2, 9), rng.uniform(-2, h + 2, 9)], axis=1).astype(np.float32)
    ref_seeds = seeds.copy()
    moved, _ = ctx.streamline(seeds, f, 2.0, 3, 100.0, variant=3)
    orc.streamline_points(ref_seeds, f, 2.0, 3, 100.0, variant=3)
    assert np.array_equal(moved.cpu().numpy(), ref_seeds)
    og = f.copy(); orc.subtract_average(og)
    assert np.abs(ctx.subtructAverage(f.copy()).cpu().numpy() - og).max() <= 2.4e-7 * max(1.0, np.abs(og).max())
    og = f.copy(); orc.stabilizer(og)
    got = ctx.stabilizer(f.copy()).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(og))          # 0/0 patch means on 1-pixel frames, as in the reference
    ok = ~np.isnan(og)
    assert np.abs(got[ok] - og[ok]).max(initial=0.0) <= 2.4e-7 * max(1.0, np.abs(og[ok]).max(initial=0.0))


def test_display_against_committed_golden_fixture(ctx):
    """tests/golden/display_64x48.npz: inputs + oracle outputs (tests/golden/make_golden.py)."""
    import os
    import torch
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "display_64x48.npz"))
    assert np.array_equal(ctx.jet_lut(), g["jet"])
    assert np.array_equal(ctx.hsv_to_bgr(g["hsv"]).cpu().numpy(), g["bgr"])


def test_host_frame_loop_pinned_double_buffer(ctx):
    """rcflow_push_frame_u8: frames from host memory through the page-locked double buffer; the flow stays on
    the device (the analysis reads it there) and equals the device-pointer loop bit for bit."""
    from ripcurrents_amd import RcflowError
    w, h, T = 333, 251, 6
    clip = synth.surf_clip(w, h, T, seed=12)
    p = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
    d = torch.as_tensor(clip).cuda()
    ref = ctx.farneback_clip(d, **p).cpu().numpy()
    ctx.stream_reset()
    with pytest.raises(RcflowError):
        ctx.stream_flow_read(w, h)                      # nothing resident yet
    padded = np.zeros((T, h, w + 19), np.uint8)         # a cv::Mat-style row step
    padded[:, :, :w] = clip
    assert ctx.push_frame_host(padded[0, :, :w], **p) is None
    for t in range(1, T):
        f = ctx.push_frame_host(padded[t, :, :w], **p)
        assert f is not None and f.shape == (h, w, 2)
        if t in (1, 3, 5):
            assert np.array_equal(ctx.stream_flow_read(w, h), ref[t - 1])
        else:
            ctx.sync()
            assert np.array_equal(f.cpu().numpy(), ref[t - 1])
    ctx.stream_reset()


def test_host_frame_loop_without_the_staging_copy(ctx):
    """rcflow_frame_buffer_acquire / rcflow_push_frame_acquired: the host produces every frame INTO the slot's page-locked
    staging buffer (the destination of the cvtColor at ripcurrents.cpp:210) and pushes it -- same flow fields as the
    device-pointer loop, the two entry points mix on one slot, and a push without an acquisition is refused."""
    from ripcurrents_amd import RcflowError
    w, h, T = 322, 241, 8
    clip = synth.surf_clip(w, h, T, seed=13)
    p = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
    ref = ctx.farneback_clip(torch.as_tensor(clip).cuda(), **p).cpu().numpy()
    ctx.stream_reset()
    with pytest.raises(RcflowError) as e:
        ctx.push_frame_acquired(w, h, **p)
    assert e.value.code == -6          # RC_ESTATE
    seen = set()
    for t in range(T):
        if t in (3, 6):                                  # the copying entry point in between
            f = ctx.push_frame_host(clip[t], **p)
        else:
            buf = ctx.frame_buffer(w, h)
            seen.add(buf.ctypes.data)
            buf[:] = clip[t]                             # "the decoder's" write
            f = ctx.push_frame_acquired(w, h, **p)
        if t == 0:
            assert f is None
        else:
            ctx.sync()
            assert np.array_equal(f.cpu().numpy(), ref[t - 1]), t
    assert len(seen) == 2                                # the two staging buffers alternate
    # an acquisition is cancelled by a copying push on the slot
    ctx.frame_buffer(w, h)
    ctx.push_frame_host(clip[0], **p)
    with pytest.raises(RcflowError):
        ctx.push_frame_acquired(w, h, **p)
    ctx.stream_reset()


def test_profile_buckets_carry_the_reference_names(ctx):
    """rcflow_profile_read_buckets: GPU time under the names of ripcurrents.cpp:103-109 / :518-524."""
    w, h = 320, 240
    clip = torch.as_tensor(synth.surf_clip(w, h, 3, seed=2)).cuda()
    ctx.analysis_reset(w, h)
    ctx.profile_reset()
    ctx.profile_enable(True)
    flow = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None)
    ctx.streamline_field(flow, 2.0, 1)
    ctx.create_histogram(flow)
    ctx.create_flow_accumulate(flow, 31)
    ctx.profile_enable(False)
    b = ctx.profile_read_buckets()
    assert list(b) == ["farneback", "polar", "threshold", "overlay", "erosion", "codec", "stream"]
    assert b["farneback"] > 0 and b["threshold"] > 0 and b["stream"] > 0 and b["polar"] == 0 and b["codec"] == 0
    ctx.profile_reset()


def test_frame_loop_on_two_streams_same_bits(ctx):
    """Option frame_overlap = 2: rcflow_push_frame_dev with the expansion of frame t + 1 on the slot's second stream
    beside the flow kernels of frame t (resident frames: complete when pushed).  Same flow fields as the one-stream
    loop, also when the caller races ahead of the GPU for many frames and across a ring wrap (chunk 3 -> 4 slots)."""
    w, h, T = 320, 240, 14
    p = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
    d = torch.as_tensor(synth.surf_clip(w, h, T, seed=21)).cuda()
    ref = ctx.farneback_clip(d, **p).cpu().numpy()
    torch.cuda.synchronize()
    for chunk in (32, 3):
        ctx.set_option("chunk", chunk)
        ctx.set_option("frame_overlap", 2)
        try:
            ctx.stream_reset()
            outs = torch.empty((T - 1, h, w, 2), dtype=torch.float32, device="cuda")
            assert ctx.push_frame(d[0], **p) is None
            for t in range(1, T):                       # no synchronisation inside the loop
                ctx.push_frame(d[t], outs[t - 1], **p)
            ctx.sync()
            torch.cuda.synchronize()
            assert np.array_equal(outs.cpu().numpy(), ref)
        finally:
            ctx.set_option("frame_overlap", 1)
            ctx.set_option("chunk", 32)
    ctx.stream_reset()


def test_two_stream_pushes_after_clip_and_pair_calls_same_bits(ctx):
    """The entry points interoperate on one slot (include/rcflow.h): two-stream frame pushes, then a clip push that reads
    every ring slot for a whole chunk, then two-stream pushes again, a pair call and a reset in between -- no host
    synchronisation anywhere.  The second stream of a frame push may only run ahead of work that earlier two-stream
    pushes recorded; after anything else it first joins the slot's stream (RcSlot::ts_streak).  Same flow fields as the
    synchronised clip call."""
    w, h, T = 640, 480, 22
    p = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
    d = torch.as_tensor(synth.surf_clip(w, h, T, seed=33)).cuda()
    ref = ctx.farneback_clip(d, **p).clone()
    torch.cuda.synchronize()
    ctx.set_option("chunk", 6)                       # ring of 7 slots: the clip push below wraps it
    ctx.set_option("frame_overlap", 2)
    try:
        for rep in range(3):
            ctx.stream_reset()
            outs = torch.zeros((T - 1, h, w, 2), dtype=torch.float32, device="cuda")
            assert ctx.push_frame(d[0], **p) is None
            for t in (1, 2, 3):                          # two-stream pushes
                ctx.push_frame(d[t], outs[t - 1], **p)
            got = ctx.push_clip(d[4:13], outs[3:12], **p)    # a clip of more than a chunk on the slot's stream
            assert got.shape[0] == 9
            for t in (13, 14, 15):                       # straight back to two-stream pushes
                ctx.push_frame(d[t], outs[t - 1], **p)
            pair = torch.zeros((h, w, 2), dtype=torch.float32, device="cuda")
            ctx.calcOpticalFlowFarneback(d[5], d[6], pair, **p)      # asynchronous pair call: restarts the stream
            assert ctx.push_frame(d[15], **p) is None
            for t in range(16, T):
                ctx.push_frame(d[t], outs[t - 1], **p)
            ctx.sync()
            torch.cuda.synchronize()
            assert torch.equal(outs, ref), "repetition %d" % rep
            assert torch.equal(pair, ref[5])
    finally:
        ctx.set_option("frame_overlap", 1)
        ctx.set_option("chunk", 32)
        ctx.stream_reset()


@pytest.mark.parametrize("use_graph", [False, True])
def test_frame_loop_step_matches_the_separate_calls(ctx, orc, use_graph):
    """rcflow_frame_loop_step: one call per frame for the whole chain of ripcurrents.cpp:194-479 (flow, streamline_field
    with the previous frame's UPPER, 250 seed streamlines, cumulative histogram + thresholds, classify / accumulate with
    the frame counter on the device, mask edges), eagerly on two streams or as one captured hipGraph launch per frame.
    Every frame's state equals what the separate calls give; framecount crosses 30 (the accumulator starts), so the
    device-side counter is exercised on both sides of that branch."""
    w, h, T = 320, 240, 36
    clip = synth.surf_clip(w, h, T, seed=44)
    seeds0 = torch.rand((250, 2), generator=torch.Generator().manual_seed(3)) * torch.tensor([w, h])
    p = dict(pyr_scale=0.5, levels=2, winsize=3, poly_n=15, poly_sigma=1.2, flags=0)

    def run(step_api):
        ctx.stream_reset()
        ctx.analysis_reset(w, h)
        seeds = seeds0.clone().float().cuda()
        mask = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
        edges = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
        snaps = []
        fc = 0
        for t in range(T):
            ctx.frame_buffer(w, h)[:] = clip[t]
            if step_api:
                f = ctx.frame_loop_step(w, h, seeds=seeds, outmask=mask, edges=edges, use_graph=use_graph, **p)
            else:
                f = ctx.push_frame_acquired(w, h, iterations=2, **p)
                if f is not None:
                    fc += 1
                    ctx.streamline_field(f, 2.0, 1)
                    ctx.streamline(seeds, f, 2.0, 1, 100.0, variant=3)
                    ctx.histogram_accumulate(f)
                    ctx.thresholds()
                    mask.copy_(ctx.create_flow_accumulate(f, fc, want=("outmask",))["outmask"])
                    edges.copy_(ctx.create_edges(mask))
            if f is not None and t in (1, 2, 17, 30, 31, 32, T - 1):
                ctx.sync()
                snaps.append((f.cpu().numpy().copy(), mask.cpu().numpy().copy(), edges.cpu().numpy().copy(),
                              seeds.cpu().numpy().copy(), ctx.histogram_words().cpu().numpy().copy(), ctx.accumulator(w, h).copy()))
        ctx.sync()
        return snaps

    a, b = run(False), run(True)
    assert len(a) == len(b) == 7
    for i, (x, y) in enumerate(zip(a, b)):
        for j, name in enumerate(("flow", "outmask", "edges", "seeds", "histogram", "accumulator")):
            assert np.array_equal(x[j], y[j]), "snapshot %d: %s" % (i, name)
    assert a[-1][5].max() > 0            # the accumulator did start (framecount > 30)
    ctx.stream_reset()


@pytest.mark.parametrize("use_graph", [False, True])
def test_frame_loop_step_refuses_to_wrap_the_histogram(ctx, use_graph):
    """The reference's `int` histogram counters (ripcurrents.cpp:147-150) wrap after 2^31 pixels -- 1035 frames of 1080p.
    rcflow_frame_loop_step refuses the frame that would (RC_ESTATE) BEFORE consuming or launching anything: the acquired
    frame buffer is still the caller's, the stream is intact, and after rcflow_histogram_reset_dev the same call goes
    through and continues the flow from the frame before."""
    w, h = 1920, 1080
    limit = (2 ** 31 - 1) // (w * h)                 # 1035 flow fields fit
    clip = synth.surf_clip(w, h, 3, seed=9)
    ctx.stream_reset()
    ctx.analysis_reset(w, h)
    mask = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    p = dict(pyr_scale=0.5, levels=2, winsize=3, poly_n=15, poly_sigma=1.2, flags=0)
    order = [0, 1, 2, 1]
    n = 0
    for t in range(limit + 1):                       # the priming frame + `limit` flows
        ctx.frame_buffer(w, h)[:] = clip[order[t % 4]]
        f = ctx.frame_loop_step(w, h, outmask=mask, use_graph=use_graph, **p)
        n += f is not None
    assert n == limit
    ctx.sync()
    st = ctx.histogram_read()
    assert 0 < st.histsum <= limit * w * h
    ctx.frame_buffer(w, h)[:] = clip[order[(limit + 1) % 4]]
    with pytest.raises(RcflowError) as e:
        ctx.frame_loop_step(w, h, outmask=mask, use_graph=use_graph, **p)
    assert e.value.code == -6                         # RC_ESTATE, nothing consumed
    assert ctx.histogram_read().histsum == st.histsum
    ctx.histogram_reset()
    f = ctx.frame_loop_step(w, h, outmask=mask, use_graph=use_graph, **p)      # the same acquired frame goes through now
    ctx.sync()
    ref = ctx.calcOpticalFlowFarneback(clip[order[limit % 4]], clip[order[(limit + 1) % 4]], None, iterations=2, **p)
    assert f is not None and np.array_equal(f.cpu().numpy(), ref)
    assert 0 < ctx.histogram_read().histsum <= w * h
    ctx.stream_reset()
    ctx.analysis_reset(64, 48)


def test_host_frame_loop_argument_and_state_errors(ctx):
    """Error behaviour of the host-frame entry points: bad arguments, oversize frames, reading a flow that does not
    exist yet, a size change in mid-stream (re-primes, like the device-pointer loop)."""
    import ctypes as C
    from ripcurrents_amd import RcflowError
    from ripcurrents_amd._lib import FarnebackParams
    lib, h = ctx._lib, ctx._h
    p = FarnebackParams(0.5, 2, 3, 2, 15, 1.2, 0)
    a = np.zeros((64, 80), np.uint8)
    assert lib.rcflow_push_frame_u8(h, 0, None, 80, 80, 64, C.byref(p)) == -1
    assert lib.rcflow_push_frame_u8(h, 0, a.ctypes.data, 40, 80, 64, C.byref(p)) == -1          # step < width
    assert lib.rcflow_push_frame_u8(h, 99, a.ctypes.data, 80, 80, 64, C.byref(p)) == -1         # no such slot
    big = np.zeros((2200, 4000), np.uint8)
    assert lib.rcflow_push_frame_u8(h, 0, big.ctypes.data, 4000, 4000, 2200, C.byref(p)) == -5  # RC_ESIZE
    ctx.stream_reset()
    with pytest.raises(RcflowError) as e:
        ctx.stream_flow_read(80, 64)
    assert e.value.code == -6
    clip = synth.surf_clip(160, 120, 3, seed=1)
    small = synth.surf_clip(96, 80, 2, seed=2)
    pk = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
    assert ctx.push_frame_host(clip[0], **pk) is None
    assert ctx.push_frame_host(clip[1], **pk) is not None
    assert ctx.push_frame_host(small[0], **pk) is None                                           # other size: primes again
    with pytest.raises(RcflowError):
        ctx.stream_flow_read(96, 80)
    f = ctx.push_frame_host(small[1], **pk)
    ctx.sync()
    got = f.cpu().numpy().copy()          # `f` aliases the slot's staging field, which the host-pointer pair call below reuses
    ref = ctx.calcOpticalFlowFarneback(small[0], small[1], None, **pk)
    assert np.array_equal(got, ref)
    with pytest.raises(RcflowError):      # the pair call restarted the stream: it has no flow field of its own any more
        ctx.stream_flow_read(96, 80)
    ctx.stream_reset()
