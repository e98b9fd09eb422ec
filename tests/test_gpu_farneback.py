"""GPU parity tests for the Farneback path (A1-A7): librcflow (HIP, through the C ABI) vs the
CPU oracle on the same seeded inputs.

Tolerances of the FAST path; the exact path (option "exact", the default for the
near-pointwise windows of main.cpp:264) is bit-identical to the oracle, see tests/test_gpu_exact.py:
  * pyramid level (A1):           bit-exact (same operation order, no contraction)
  * polynomial expansion (A2):    |dR| <= 2e-4 absolute on coefficients of O(1..100)
                                  (oracle: float vertical + double horizontal sums; HIP: fp32
                                  sums on DC-removed data + double epilogue)
  * one flow iteration (A3-A6):   |dflow| <= 1e-3 px on >= 99.9 % of pixels, given identical R
  * end to end:                   tests/_parity.py: max-abs and p99.9 of |dflow| over the pixels whose final
                                  2x2 system has det > 1e-2 (p99.9 <= 1e-3 px, max <= 5e-3 px) and
                                  max <= 5e-2 px over all other pixels -- 8(d)'s metric, with caps.
"""
import numpy as np
import pytest
import torch

from ripcurrents_amd import synth
from _parity import assert_conditioned

pytestmark = pytest.mark.gpu

RC215 = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
MAIN264 = dict(RC215, flags=256)
MAIN609 = dict(RC215, winsize=20, iterations=3, flags=256)
MAIN1119 = dict(RC215, winsize=10, iterations=3, flags=256)
AND167 = dict(RC215, levels=3, winsize=5, iterations=3)


def _oracle_flow(orc, a, b, p):
    return orc.farneback(a, b, p["pyr_scale"], p["levels"], p["winsize"], p["iterations"], p["poly_n"],
                         p["poly_sigma"], p["flags"])


def _oracle_diag(orc, a, b, p, nthreads=4):
    """(flow, det of the final solve per pixel) -- the conditioning SURVEY 8(d) states its tolerance on"""
    ref, det_last, det_min = orc.farneback_diag(a, b, p["pyr_scale"], p["levels"], p["winsize"], p["iterations"], p["poly_n"],
                                                p["poly_sigma"], p["flags"], nthreads=nthreads)
    return ref, (det_last, det_min)


def _report(name, got, ref):
    err = np.abs(got - ref).max(-1)
    stats = dict(max=float(err.max()), p50=float(np.percentile(err, 50)), p99=float(np.percentile(err, 99)),
                 p999=float(np.percentile(err, 99.9)), frac_1e3=float((err <= 1e-3).mean()))
    print("\n[parity] %s: %s" % (name, stats))
    return stats


@pytest.mark.parametrize("size,k", [((640, 480), 0), ((640, 480), 1), ((640, 480), 2), ((333, 251), 1),
                                    ((1920, 1080), 2), ((97, 65), 0), ((1920, 1080), 3), ((1920, 1080), 4),
                                    ((3840, 2160), 4), ((1000, 700), 3), ((333, 251), 2), ((2001, 1127), 1),
                                    ((1920, 1080), 0), ((333, 251), 0), ((2001, 1127), 0), ((6, 5), 0)])
def test_pyr_level_bit_exact(ctx, orc, size, k):
    w, h = size
    img = synth.surf_clip(w, h, 1, seed=7)[0]
    g = orc.level_geometry(w, h, 0.5, 8, k)
    ref = orc.pyr_level(img, g["sigma"], g["ksize"], g["w"], g["h"])
    got = ctx.stage_pyr_level(img, 0.5, k).cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), "max diff %g" % np.abs(got - ref).max()
    # the earlier kernels (per-pixel direct form / LDS-staged region) stay selectable and agree
    ctx.set_option("ablate", 4096)
    try:
        old = ctx.stage_pyr_level(img, 0.5, k).cpu().numpy()
    finally:
        ctx.set_option("ablate", 0)
    assert np.array_equal(old, ref)


def test_pyr_level_noninteger_scale(ctx, orc):
    w, h = 500, 375
    img = synth.surf_clip(w, h, 1, seed=3)[0]
    for k in (1, 2, 3):
        g = orc.level_geometry(w, h, 0.8, 8, k)
        ref = orc.pyr_level(img, g["sigma"], g["ksize"], g["w"], g["h"])
        got = ctx.stage_pyr_level(img, 0.8, k).cpu().numpy()
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("size", [(640, 480), (130, 70), (64, 32), (1920, 1080)])
@pytest.mark.parametrize("n,sigma", [(15, 1.2), (5, 1.1), (7, 1.5)])
def test_polyexp(ctx, orc, size, n, sigma):
    w, h = size
    if (w, h) == (1920, 1080) and n != 15:
        pytest.skip("full size only for the reference's parameters")
    img = synth.surf_clip(w, h, 1, seed=11)[0]
    I = orc.pyr_level(img, 0.0, 3, w, h)
    ref = orc.polyexp(I, n, sigma)
    got = ctx.stage_polyexp(I, n, sigma).cpu().numpy()
    err = np.abs(got - ref)
    print("\n[parity] polyexp %dx%d n=%d: max %g, rel-to-max %g" % (w, h, n, err.max(), err.max() / np.abs(ref).max()))
    assert err.max() <= 2e-4


def test_polyexp_matrix_core_vertical_pass(ctx, orc):
    """Option poly_mfma: the vertical pass as a banded Toeplitz product on v_mfma_f32_16x16x4_f32
    (exact f32 fma chain).  Same tolerance as the VALU form; it is off by default because it measured
    25 % slower (the band wastes half of every 16 x 32 operand)."""
    w, h = 640, 480
    I = (synth.surf_clip(w, h, 1, seed=2)[0].astype(np.float32)) * 0.7 + 3.0
    ref = orc.polyexp(I, 15, 1.2)
    ctx.set_option("poly_mfma", 1)
    try:
        got = ctx.stage_polyexp(I, 15, 1.2).cpu().numpy()
    finally:
        ctx.set_option("poly_mfma", 0)
    assert np.abs(got - ref).max() <= 2e-4


def test_polyexp_exact_taps_option(ctx, orc):
    """Dropping the negligible taps (default) vs evaluating all 31 moves R by < 1e-5."""
    img = synth.surf_clip(320, 240, 1, seed=5)[0]
    I = orc.pyr_level(img, 0.0, 3, 320, 240)
    a = ctx.stage_polyexp(I, 15, 1.2).cpu().numpy()
    ctx.set_option("exact_taps", 1)
    try:
        b = ctx.stage_polyexp(I, 15, 1.2).cpu().numpy()
    finally:
        ctx.set_option("exact_taps", 0)
    ref = orc.polyexp(I, 15, 1.2)
    assert np.abs(a - b).max() < 1e-5
    assert np.abs(b - ref).max() <= 2e-4


def test_polyexp_quadratic_known_answer(ctx):
    """Analytic KAT: for I = a + b x + c y + d x^2 + e y^2 + f xy the interior output is
    (c', b', e, d, f) with first-order terms evaluated at the pixel."""
    h, w = 96, 128
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    a, b, c, d, e, f = 20.0, 0.5, -0.25, 0.01, -0.02, 0.015
    I = (a + b * xs + c * ys + d * xs * xs + e * ys * ys + f * xs * ys).astype(np.float32)
    R = ctx.stage_polyexp(I, 15, 1.2).cpu().numpy()
    m = 20
    Ri = R[m:-m, m:-m].astype(np.float64)
    xi, yi = xs[m:-m, m:-m], ys[m:-m, m:-m]
    assert np.abs(Ri[..., 0] - (c + 2 * e * yi + f * xi)).max() < 2e-3
    assert np.abs(Ri[..., 1] - (b + 2 * d * xi + f * yi)).max() < 2e-3
    assert np.abs(Ri[..., 2] - e).max() < 2e-3
    assert np.abs(Ri[..., 3] - d).max() < 2e-3
    assert np.abs(Ri[..., 4] - f).max() < 2e-3


@pytest.mark.parametrize("winsize,flags", [(3, 0), (3, 256), (10, 256), (20, 256), (5, 0), (4, 0)])
def test_flow_iteration_stage(ctx, orc, winsize, flags):
    """One UpdateMatrices + window + solve on identical R0, R1, flow_in."""
    w, h = 320, 240
    clip = synth.surf_clip(w, h, 2, seed=21)
    I0 = orc.pyr_level(clip[0], 0.0, 3, w, h)
    I1 = orc.pyr_level(clip[1], 0.0, 3, w, h)
    R0, R1 = orc.polyexp(I0), orc.polyexp(I1)
    rng = np.random.RandomState(1)
    fin = (rng.randn(h, w, 2) * 1.5).astype(np.float32)
    M = orc.update_matrices(R0, R1, fin)
    ref = fin.copy()
    orc.update_flow(R0, R1, ref, M, winsize, False, bool(flags & 256))
    got = ctx.stage_flow_iter(R0, R1, fin, winsize, flags).cpu().numpy()
    st = _report("flow_iter win=%d flags=%d" % (winsize, flags), got, ref)
    assert st["frac_1e3"] >= 0.999
    # zero initial flow = the coarsest-level case
    M = orc.update_matrices(R0, R1, np.zeros_like(fin))
    ref0 = np.zeros_like(fin)
    orc.update_flow(R0, R1, ref0, M, winsize, False, bool(flags & 256))
    got0 = ctx.stage_flow_iter(R0, R1, None, winsize, flags).cpu().numpy()
    assert (np.abs(got0 - ref0).max(-1) <= 1e-3).mean() >= 0.999


@pytest.mark.parametrize("winsize", [10, 20])
def test_flow_iteration_stage_sweep_kernel(ctx, orc, winsize):
    """The strip-sweep kernel (large launches of the Gaussian winsize 10 / 20 call sites) forced on a
    small stage input: same oracle comparison as the tile kernel, and the same bits as the tile kernel."""
    w, h = 333, 217
    clip = synth.surf_clip(w, h, 2, seed=22)
    I0 = orc.pyr_level(clip[0], 0.0, 3, w, h)
    I1 = orc.pyr_level(clip[1], 0.0, 3, w, h)
    R0, R1 = orc.polyexp(I0), orc.polyexp(I1)
    fin = (np.random.RandomState(2).randn(h, w, 2) * 1.5).astype(np.float32)
    M = orc.update_matrices(R0, R1, fin)
    ref = fin.copy()
    orc.update_flow(R0, R1, ref, M, winsize, False, True)
    try:
        ctx.set_option("ablate", 65536)
        tile = ctx.stage_flow_iter(R0, R1, fin, winsize, 256).cpu().numpy()
        ctx.set_option("ablate", 8388608)
        got = ctx.stage_flow_iter(R0, R1, fin, winsize, 256).cpu().numpy()
    finally:
        ctx.set_option("ablate", 0)
    assert _report("flow_iter sweep win=%d" % winsize, got, ref)["frac_1e3"] >= 0.999
    assert np.array_equal(got, tile)


@pytest.mark.parametrize("size,npairs", [((333, 251), 3), ((97, 70), 2), ((64, 19), 1), ((1000, 9), 1)])
def test_sweep_kernel_whole_path_identical_to_tile_kernel(ctx, size, npairs):
    """MAIN609 / MAIN1119 through farneback_clip with either kernel forced: identical flow fields
    (ragged strips, segments shorter than the window, images lower than the window radius)."""
    w, h = size
    clip = synth.surf_clip(w, h, npairs + 1, seed=9)
    try:
        for p in (MAIN609, MAIN1119):
            ctx.set_option("ablate", 65536)
            a = ctx.farneback_clip(clip, **p).cpu().numpy()
            ctx.set_option("ablate", 8388608)
            b = ctx.farneback_clip(clip, **p).cpu().numpy()
            assert np.isfinite(a).all() and np.array_equal(a, b)
    finally:
        ctx.set_option("ablate", 0)


@pytest.mark.parametrize("name,p,size", [
    ("RC215 640x480", RC215, (640, 480)),
    ("MAIN609 win20", MAIN609, (640, 480)),
    ("MAIN1119 win10", MAIN1119, (640, 480)),
    ("AND167 levels3 win5", AND167, (640, 480)),
    ("ragged 333x251", RC215, (333, 251)),
    ("tiny 40x36 (levels cropped)", RC215, (40, 36)),
])
def test_end_to_end_parity(ctx, orc, name, p, size):
    """Fast path against the oracle with SURVEY 8(d)'s determinant-conditioned metric (tests/_parity.py)."""
    w, h = size
    clip = synth.surf_clip(w, h, 2, seed=1234)
    ref, det_last = _oracle_diag(orc, clip[0], clip[1], p)
    got = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **p)
    st = assert_conditioned(name, got, ref, *det_last)
    assert st["p50"] <= 1e-4


def test_end_to_end_main264_is_the_oracle_bit_for_bit(ctx, orc):
    """main.cpp:264, :742, ripcurrents_module.cpp:712, main_old.cpp:324 (Gaussian winsize 3): the library runs
    upstream's operation order for this near-pointwise window by default."""
    for size in ((640, 480), (333, 251)):
        clip = synth.surf_clip(size[0], size[1], 2, seed=1234)
        ref = _oracle_flow(orc, clip[0], clip[1], MAIN264)
        assert np.array_equal(ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **MAIN264), ref)


@pytest.mark.parametrize("size", [(32, 32), (33, 47), (64, 33), (100, 37), (130, 70), (257, 129), (31, 64), (1, 1), (2, 200)])
def test_ragged_and_tiny_sizes(ctx, orc, size):
    """Edge sizes (every kernel's border path, level cropping at min_size = 32, images smaller than a
    tile / a window): same answers as the oracle, for every kernel family."""
    w, h = size
    clip = synth.surf_clip(max(w, 8), max(h, 8), 2, seed=17)[:, :h, :w].copy()
    for p in (RC215, MAIN264, dict(RC215, levels=5, iterations=3), dict(RC215, winsize=7, iterations=1), AND167, MAIN1119):
        ref, det_last = _oracle_diag(orc, clip[0], clip[1], p, nthreads=1)
        got = np.asarray(ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **p))
        assert got.shape == ref.shape
        if p is MAIN264:
            assert np.array_equal(got, ref), size        # exact path by default
        else:
            assert_conditioned("%dx%d win%d" % (w, h, p["winsize"]), got, ref, *det_last, tier="stress")


def test_device_entry_point_matches_host_entry_point(ctx):
    clip = synth.surf_clip(320, 240, 2, seed=9)
    a = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **RC215)
    d = torch.as_tensor(clip).cuda()
    b = ctx.calcOpticalFlowFarneback(d[0], d[1], None, **RC215)
    ctx.sync()
    assert np.array_equal(a, b.cpu().numpy())


def test_strided_buffers(ctx):
    """cv::Mat-style row steps on inputs and output."""
    clip = synth.surf_clip(300, 200, 2, seed=2)
    big = np.zeros((2, 200, 352), np.uint8)
    big[:, :, :300] = clip
    flow_big = np.full((200, 320, 2), np.nan, np.float32)
    a = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **RC215)
    ctx.calcOpticalFlowFarneback(big[0, :, :300], big[1, :, :300], flow_big[:, :300], **RC215)
    assert np.array_equal(flow_big[:, :300], a)
    assert np.isnan(flow_big[:, 300:]).all()


def test_zero_motion_and_translation(ctx):
    """Analytic KATs: identical frames give exactly zero flow away from
    the last rows/columns; a translated texture recovers the translation."""
    clip = synth.translating_clip(640, 480, 2, u=1.25, v=-0.75)
    z = ctx.calcOpticalFlowFarneback(clip[0], clip[0], None, **RC215)
    assert np.abs(z[:-24, :-24]).max() == 0.0
    f = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **RC215)
    inner = f[40:-40, 40:-40]
    assert abs(np.median(inner[..., 0]) - 1.25) < 0.08
    assert abs(np.median(inner[..., 1]) + 0.75) < 0.08


def test_clip_and_streaming_match_pairwise(ctx):
    """Clip batching (chunks of pairs per launch) and the streaming push_frame path give
    bit-identical flows to the two-image call."""
    T, w, h = 7, 320, 240
    clip = synth.surf_clip(w, h, T, seed=77)
    d = torch.as_tensor(clip).cuda()
    flows = ctx.farneback_clip(d, **RC215)
    ctx.sync()
    flows = flows.cpu().numpy()
    for t in range(T - 1):
        ref = ctx.calcOpticalFlowFarneback(clip[t], clip[t + 1], None, **RC215)
        assert np.array_equal(flows[t], ref), "pair %d" % t
    ctx.stream_reset()
    outs = []
    for t in range(T):
        r = ctx.push_frame(d[t], **RC215)
        if r is not None:
            ctx.sync()
            outs.append(r.cpu().numpy())
    assert len(outs) == T - 1
    for t in range(T - 1):
        assert np.array_equal(outs[t], flows[t])


def test_two_stream_overlap_option_same_bits(ctx):
    """Option overlap: expansions of chunk c+1 on a second stream beside the flow kernels of chunk c
    (event-joined).  Same results; off by default because it measured 1 % slower."""
    clip = torch.as_tensor(synth.surf_clip(320, 240, 12, seed=8)).cuda()
    out = torch.empty((11, 240, 320, 2), dtype=torch.float32, device="cuda")
    ctx.set_option("chunk", 4)
    try:
        ctx.farneback_clip(clip, out, **RC215)
        a = out.cpu().numpy().copy()
        ctx.set_option("overlap", 1)
        out.zero_()
        ctx.farneback_clip(clip, out, **RC215)
        ctx.sync()
        b = out.cpu().numpy()
    finally:
        ctx.set_option("overlap", 0)
        ctx.set_option("chunk", 32)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("size,levels,n", [((640, 480), 2, 5), ((332, 252), 2, 4), ((328, 244), 1, 4), ((1024, 512), 4, 4),
                                           ((64, 32), 2, 4), ((8, 8), 1, 4), ((72, 40), 2, 4), ((333, 251), 2, 4)])
def test_fused_pyramid_same_bits(ctx, size, levels, n):
    """Clips of more than two frames at pyr_scale 0.5 with exact half / quarter sizes: pyramid scales 1
    and 2 come out of the scale-0 expansion launch (option fuse_pyr).  Same flow bits as with the
    pyramid kernels (which test_pyr_level_bit_exact pins to the oracle); sizes that are not exact
    multiples (333x251) take the pyramid kernels either way."""
    w, h = size
    clip = synth.surf_clip(w, h, n, seed=3)
    p = dict(RC215, levels=levels)
    try:
        ctx.set_option("fuse_pyr", 0)
        a = ctx.farneback_clip(clip, **p).cpu().numpy()
        ctx.set_option("fuse_pyr", 1)
        b = ctx.farneback_clip(clip, **p).cpu().numpy()
    finally:
        ctx.set_option("fuse_pyr", 1)
    assert np.isfinite(a).all() and np.array_equal(a, b)


def test_chunk_option_invariance(ctx):
    clip = torch.as_tensor(synth.surf_clip(256, 192, 6, seed=4)).cuda()
    a = ctx.farneback_clip(clip, **RC215).cpu().numpy()
    for c in (1, 2, 5):
        ctx.set_option("chunk", c)
        b = ctx.farneback_clip(clip, **RC215).cpu().numpy()
        assert np.array_equal(a, b)
    ctx.set_option("chunk", 32)


@pytest.mark.parametrize("size,params,T", [((640, 480), dict(levels=2), 10), ((333, 251), dict(levels=2, flags=256), 7),
                                           ((1024, 576), dict(levels=4), 6), ((700, 500), dict(pyr_scale=0.7, levels=3), 6),
                                           ((257, 130), dict(levels=0, iterations=4), 12), ((1920, 1080), dict(levels=2), 10)])
def test_tile_chains_same_bits(ctx, size, params, T):
    """Option `chain`: a block of the fused winsize-3 kernel walks consecutive pairs on its tile and takes pair z + 1's R0
    out of pair z's R1 window in LDS (k_flow_iter2_rrc; the reference's u_f1.copyTo(u_f2), ripcurrents.cpp:194-221, at tile
    level).  Same bits as one pair per block for every chain length, border tiles, every flow_in form (zeros, previous
    iteration, coarser scale at exact and inexact ratios), chain groups of unequal length and chunk ends (chunk 5 of 9 pairs)."""
    w, h = size
    p = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
    p.update(params)
    clip = torch.as_tensor(synth.surf_clip(w, h, T, seed=11)).cuda()
    ctx.set_option("exact", 0)
    try:
        ctx.set_option("chain", 1)
        a = ctx.farneback_clip(clip, **p).clone()
        assert torch.isfinite(a).all()
        ctx.set_option("ablate", 33554432)       # RC_ABL_FORCE_CHAIN: chains whatever the launch size
        for chain, chunk in ((2, 32), (3, 32), (4, 5), (8, 32), (64, 32)):
            ctx.set_option("chain", chain)
            ctx.set_option("chunk", chunk)
            b = ctx.farneback_clip(clip, **p).clone()
            assert torch.equal(a, b), "chain %d chunk %d" % (chain, chunk)
        # the streaming entry point (a continuing segment: the slot ring wraps between calls)
        ctx.set_option("chain", 4)
        ctx.set_option("chunk", 4)
        ctx.stream_reset()
        out = torch.empty((T - 1, h, w, 2), dtype=torch.float32, device="cuda")
        ctx.push_clip(clip[0:1], out, **p)
        got = ctx.push_clip(clip[1:], out, **p)
        ctx.sync()
        assert got.shape[0] == T - 1 and torch.equal(a, got)
    finally:
        ctx.set_option("ablate", 0)
        ctx.set_option("chain", 8)
        ctx.set_option("chunk", 32)
        ctx.set_option("exact", -1)
        ctx.stream_reset()


def test_full_size_1080p_parity(ctx, orc):
    """BASELINE config 2 at full size against the oracle (one pair, ~2 s of CPU)."""
    clip = synth.surf_clip(1920, 1080, 2, seed=1234)
    ref, det_last = _oracle_diag(orc, clip[0], clip[1], RC215, nthreads=8)
    got = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **RC215)
    st = assert_conditioned("C2 1080p RC215", got, ref, *det_last)
    assert st["frac_1e3"] >= 0.999 and st["p50"] <= 1e-4


@pytest.mark.parametrize("name,p", [("MAIN1119 gaussian win10 it3", MAIN1119), ("MAIN609 gaussian win20 it3", MAIN609),
                                    ("AND167 box win5 it3 levels3", AND167)])
def test_full_size_1080p_parity_other_call_sites(ctx, orc, name, p):
    """The other reference parameter sets at 1080p against the oracle (8 oracle threads, a few seconds each)."""
    clip = synth.surf_clip(1920, 1080, 2, seed=77)
    ref, det_last = _oracle_diag(orc, clip[0], clip[1], p, nthreads=8)
    got = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **p)
    st = assert_conditioned("1080p " + name, got, ref, *det_last)
    assert st["p50"] <= 1e-4


def test_4k_five_scales_properties(ctx):
    """BASELINE config 3 (3840x2160, levels=4): size-independent properties instead of the
    oracle: zero motion -> zero interior flow; flipping both frames left-right mirrors the flow."""
    p = dict(RC215, levels=4)
    clip = synth.surf_clip(3840, 2160, 2, seed=5)
    d = torch.as_tensor(clip).cuda()
    z = ctx.calcOpticalFlowFarneback(d[0], d[0], None, **p)
    ctx.sync()
    assert float(z[:-200, :-200].abs().max()) == 0.0
    f = ctx.calcOpticalFlowFarneback(d[0], d[1], None, **p).clone()
    fm = ctx.calcOpticalFlowFarneback(d[0].flip(1).contiguous(), d[1].flip(1).contiguous(), None, **p)
    ctx.sync()
    fm = fm.flip(1)
    core = (slice(100, -100), slice(100, -100))
    ex = (f[..., 0][core] + fm[..., 0][core]).abs()
    ey = (f[..., 1][core] - fm[..., 1][core]).abs()
    # mirror symmetry is broken only by the one-sided border rule (x1 < w-1) and rounding
    assert float((ex < 1e-2).float().mean()) > 0.99 and float((ey < 1e-2).float().mean()) > 0.99


def test_4k_config3_advection_on_gpu_flow(ctx, orc):
    """BASELINE config 3's second half: 250 seed streamlines (ripcurrents.cpp:170-172, :283-285) and 5
    streaklines (main.cpp:118) advected through the 4K flow field the GPU produced -- the sampler is
    the reference's own arithmetic, so positions must match the oracle bit for bit on that field."""
    from ripcurrents_amd.api import Streakline
    p = dict(RC215, levels=4)
    w, h = 3840, 2160
    clip = torch.as_tensor(synth.surf_clip(w, h, 3, seed=9)).cuda()
    rng = np.random.RandomState(1)
    seeds = np.stack([rng.uniform(0, w, 250), rng.uniform(0, h, 250)], axis=1).astype(np.float32)
    ref_seeds = seeds.copy()
    gens = [(w * (0.2 + 0.15 * i), h * 0.5) for i in range(5)]
    sls = [Streakline(g) for g in gens]
    overts = [np.zeros((8, 2), np.float32) for _ in gens]
    state = [[1, 1] for _ in gens]
    for i, g in enumerate(gens):
        overts[i][0] = g
    for t in range(2):
        flow = ctx.calcOpticalFlowFarneback(clip[t], clip[t + 1], None, **p)
        hflow = flow.cpu().numpy()
        moved, _ = ctx.streamline(seeds, flow, 2.0, 1, 100.0, variant=3)
        seeds = moved.cpu().numpy()
        orc.streamline_points(ref_seeds, hflow, 2.0, 1, 100.0, variant=3)
        assert np.array_equal(seeds, ref_seeds)
        for i, sl in enumerate(sls):
            sl.run(ctx, flow, w, h)
            state[i] = list(orc.streakline_step(overts[i], state[i][0], gens[i], hflow, 1.0, state[i][1]))
            assert sl.numberOfVertices == state[i][0] and sl.frameCount == state[i][1]
            assert np.array_equal(np.asarray(sl.vertices, np.float32), overts[i][:state[i][0]])


SWEEP = [
    dict(pyr_scale=0.8, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=1.1, flags=0),      # OpenCV sample defaults
    dict(pyr_scale=0.8, levels=3, winsize=15, iterations=3, poly_n=7, poly_sigma=1.5, flags=256),
    dict(pyr_scale=0.75, levels=4, winsize=7, iterations=2, poly_n=7, poly_sigma=1.5, flags=0),
    dict(pyr_scale=0.6, levels=5, winsize=9, iterations=1, poly_n=5, poly_sigma=1.1, flags=256),
    dict(pyr_scale=0.5, levels=1, winsize=25, iterations=4, poly_n=15, poly_sigma=1.2, flags=256),
    dict(pyr_scale=0.5, levels=0, winsize=3, iterations=1, poly_n=15, poly_sigma=1.2, flags=0),      # one scale, one iteration
    dict(pyr_scale=0.5, levels=3, winsize=4, iterations=3, poly_n=5, poly_sigma=1.1, flags=0),       # even window
    dict(pyr_scale=0.3, levels=2, winsize=11, iterations=2, poly_n=7, poly_sigma=1.5, flags=0),
    dict(pyr_scale=0.9, levels=6, winsize=5, iterations=2, poly_n=5, poly_sigma=1.1, flags=256),
    dict(pyr_scale=0.5, levels=2, winsize=21, iterations=2, poly_n=10, poly_sigma=2.0, flags=0),
]


@pytest.mark.parametrize("i", range(len(SWEEP)))
def test_parameter_sweep_beyond_the_reference_call_sites(ctx, orc, i):
    """The drop-in signature takes any parameter set, not only the five the reference uses: other
    pyramid ratios (non-integer level sizes), level counts, window sizes (generic kernels), expansion
    sizes and iteration counts, against the oracle at 230x170."""
    p = SWEEP[i]
    clip = synth.surf_clip(230, 170, 2, seed=100 + i)
    ref, det_last = _oracle_diag(orc, clip[0], clip[1], p)
    got = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **p)
    st = assert_conditioned("sweep %d %s" % (i, p), got, ref, *det_last)
    assert st["p50"] <= 2e-4


@pytest.mark.parametrize("seed", range(12))
def test_randomised_shapes_and_parameters_default_options(ctx, orc, seed):
    """Twelve seeded draws over size (120 .. 420, even and odd), pyramid ratio and depth, window, iterations and expansion
    size through the DEFAULT options (fast kernels; the exact path where `exact` = -1 selects it), against the oracle with
    SURVEY 8(d)'s conditioned metric."""
    rng = np.random.RandomState(9000 + seed)
    w, h = int(rng.randint(120, 421)), int(rng.randint(120, 421))
    p = dict(pyr_scale=float(rng.choice([0.5, 0.5, 0.6, 0.75, 0.8])), levels=int(rng.randint(0, 4)),
             winsize=int(rng.choice([3, 3, 4, 5, 5, 7, 10, 13, 20])), iterations=int(rng.randint(1, 4)),
             poly_n=int(rng.choice([5, 7, 15])), poly_sigma=float(rng.choice([1.1, 1.2, 1.5])),
             flags=int(rng.choice([0, 256])))
    clip = synth.surf_clip(w, h, 2, seed=int(rng.randint(1 << 30)))
    ref, det_last = _oracle_diag(orc, clip[0], clip[1], p)
    got = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **p)
    assert_conditioned("random %d: %dx%d %s" % (seed, w, h, p), got, ref, *det_last, tier="config" if min(w, h) >= 200 else "stress")


def _hostile_images(w, h):
    rng = np.random.RandomState(1)
    yy, xx = np.mgrid[0:h, 0:w]
    return {
        "noise": (rng.randint(0, 256, (h, w)).astype(np.uint8), rng.randint(0, 256, (h, w)).astype(np.uint8)),
        "constant": (np.full((h, w), 77, np.uint8), np.full((h, w), 77, np.uint8)),
        "black to white": (np.zeros((h, w), np.uint8), np.full((h, w), 255, np.uint8)),
        "checker 1px": ((((xx + yy) & 1) * 255).astype(np.uint8), (((xx + yy + 1) & 1) * 255).astype(np.uint8)),
        "vertical bars": (((xx // 4 & 1) * 255).astype(np.uint8), (((xx + 2) // 4 & 1) * 255).astype(np.uint8)),
        "edge": (np.where(xx < w // 2, 0, 255).astype(np.uint8), np.where(xx < w // 2 + 3, 0, 255).astype(np.uint8)),
    }


@pytest.mark.parametrize("name", ["noise", "constant", "black to white", "checker 1px", "vertical bars", "edge"])
def test_hostile_images(ctx, orc, name):
    """Uncorrelated noise (flows of tens of pixels: gathers far outside the block's window), textureless and
    saturated frames (singular 2x2 systems) through three reference parameter sets: finite, and the oracle's values."""
    a, b = _hostile_images(320, 240)[name]
    for p in (RC215, MAIN609, AND167):
        ref, det_last = _oracle_diag(orc, a, b, p)
        got = ctx.calcOpticalFlowFarneback(a, b, None, **p)
        assert_conditioned("%s win %d" % (name, p["winsize"]), got, ref, *det_last, tier="stress" if name == "noise" else "config")


def test_noise_clip_1080p_is_finite(ctx):
    clip = torch.randint(0, 256, (5, 1080, 1920), dtype=torch.uint8, device="cuda")
    for p in (RC215, MAIN609, AND167):
        out = ctx.farneback_clip(clip, **p)
        assert bool(torch.isfinite(out).all())


def test_stream_reprimes_on_parameter_or_size_change(ctx):
    """push_frame keeps the previous frame's expansion; a call with other parameters or another frame size
    must not reuse it: it primes again (no flow), and the next flow equals a fresh two-image call."""
    clip = synth.surf_clip(320, 240, 4, seed=3)
    small = synth.surf_clip(256, 192, 3, seed=4)
    npy = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
    for p2 in (MAIN1119, dict(RC215, poly_n=5, poly_sigma=1.1), dict(RC215, levels=3), dict(RC215, pyr_scale=0.7),
               dict(RC215, iterations=3)):
        ctx.stream_reset()
        assert ctx.push_frame(clip[0], **RC215) is None
        assert ctx.push_frame(clip[1], **RC215) is not None
        assert ctx.push_frame(clip[2], **p2) is None
        got = npy(ctx.push_frame(clip[3], **p2)).copy()
        assert np.array_equal(got, npy(ctx.calcOpticalFlowFarneback(clip[2], clip[3], None, **p2)))
    ctx.stream_reset()
    ctx.push_frame(clip[0], **RC215)
    ctx.push_frame(clip[1], **RC215)
    assert ctx.push_frame(small[0], **RC215) is None
    got = npy(ctx.push_frame(small[1], **RC215)).copy()
    assert np.array_equal(got, npy(ctx.calcOpticalFlowFarneback(small[0], small[1], None, **RC215)))
    ctx.stream_reset()


@pytest.mark.parametrize("p", [RC215, MAIN1119])
def test_push_clip_continues_the_stream(ctx, p):
    """A segment pushed in uneven batches (rcflow_push_clip_dev), mixed with single push_frame calls, gives the
    flow fields of one clip call over the whole segment: the stream carries over, every frame is expanded once."""
    clip = torch.as_tensor(synth.surf_clip(200, 150, 12, seed=6)).cuda()
    whole = ctx.farneback_clip(clip, **p).cpu().numpy()                      # 11 fields
    for chunk in (32, 2):
        ctx.set_option("chunk", chunk)
        ctx.stream_reset()
        got = []
        a = ctx.push_clip(clip[0:4], **p)                 # primes: 3 fields
        assert a.shape[0] == 3
        got += [a.cpu().numpy()]
        b = ctx.push_clip(clip[4:5], **p)                 # one frame: 1 field
        assert b.shape[0] == 1
        got += [b.cpu().numpy()]
        c = ctx.push_frame(clip[5], **p)                  # the single-frame call continues the same stream
        got += [c.cpu().numpy()[None]]
        d = ctx.push_clip(clip[6:12], **p)                # 6 fields
        assert d.shape[0] == 6
        got += [d.cpu().numpy()]
        assert np.array_equal(np.concatenate(got), whole)
    ctx.set_option("chunk", 32)
    # a first call of one frame only primes
    ctx.stream_reset()
    assert ctx.push_clip(clip[0:1], **p).shape[0] == 0
    assert np.array_equal(ctx.push_clip(clip[1:3], **p).cpu().numpy(), whole[:2])
    # other parameters: primes again
    q = dict(p, iterations=p["iterations"] + 1)
    assert ctx.push_clip(clip[3:6], **q).shape[0] == 2
    ctx.stream_reset()


def test_error_codes(ctx):
    from ripcurrents_amd import RcflowError
    a = np.zeros((64, 64), np.uint8)
    with pytest.raises(RcflowError) as e:
        ctx.calcOpticalFlowFarneback(a, a, None, 1.5, 2, 3, 2, 15, 1.2, 0)      # pyr_scale >= 1
    assert e.value.code == -1
    with pytest.raises(RcflowError) as e:
        ctx.calcOpticalFlowFarneback(a, a, None, 0.5, 2, 3, 2, 15, 1.2, 4)      # USE_INITIAL_FLOW
    assert e.value.code == -1
    big = np.zeros((2200, 4000), np.uint8)
    with pytest.raises(RcflowError) as e:
        ctx.calcOpticalFlowFarneback(big, big, None, **RC215)
    assert e.value.code == -5


@pytest.mark.parametrize("p", [RC215, MAIN264, dict(RC215, iterations=3), dict(RC215, iterations=5, levels=1)])
def test_fused_iterations_bit_identical(ctx, p):
    """Two iterations per launch (default) vs one launch per iteration: same bits."""
    clip = synth.surf_clip(333, 251, 2, seed=8)
    ctx.set_option("exact", 0)          # the fast kernels, also for the Gaussian winsize-3 set
    try:
        a = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **p)
        ctx.set_option("fuse_iters", 0)
        b = ctx.calcOpticalFlowFarneback(clip[0], clip[1], None, **p)
    finally:
        ctx.set_option("fuse_iters", 1)
        ctx.set_option("exact", -1)
    assert np.array_equal(a, b)


def test_against_committed_golden_fixtures(ctx):
    """tests/golden/*.npz (inputs + oracle outputs, made by tests/golden/make_golden.py)."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name in ("farneback_rc215_96x80.npz", "farneback_main1119_96x80.npz", "farneback_rc215_320x240.npz"):
        g = np.load(os.path.join(gold, name))
        got = ctx.calcOpticalFlowFarneback(g["prev"], g["next"], None, g["pyr_scale"].item(), g["levels"].item(),
                                           g["winsize"].item(), g["iters"].item(), g["poly_n"].item(),
                                           g["poly_sigma"].item(), g["flags"].item())
        # SURVEY 8(d)'s conditioned metric on the determinants the fixture carries (the 96x80 ones are mostly border band
        # with two cropped scales: tier "stress"; 320x240 has three full scales: tier "config")
        assert_conditioned("golden " + name, got, g["flow"], g["det_last"].astype(np.float64), g["det_min"],
                           tier="config" if "320x240" in name else "stress")
    # main.cpp:264's parameters: the default path is exact -> the committed vector bit for bit; and every parameter set
    # through option exact = 1
    g = np.load(os.path.join(gold, "farneback_main264_96x80.npz"))
    args = [g[k].item() for k in ("pyr_scale", "levels", "winsize", "iters", "poly_n", "poly_sigma", "flags")]
    assert np.array_equal(ctx.calcOpticalFlowFarneback(g["prev"], g["next"], None, *args), g["flow"])
    ctx.set_option("exact", 1)
    try:
        for name in ("farneback_rc215_96x80.npz", "farneback_main1119_96x80.npz", "farneback_rc215_320x240.npz"):
            g = np.load(os.path.join(gold, name))
            args = [g[k].item() for k in ("pyr_scale", "levels", "winsize", "iters", "poly_n", "poly_sigma", "flags")]
            assert np.array_equal(ctx.calcOpticalFlowFarneback(g["prev"], g["next"], None, *args), g["flow"]), name
    finally:
        ctx.set_option("exact", -1)
    from ripcurrents_amd.api import HistState
    h = np.load(os.path.join(gold, "histogram_96x80.npz"))
    st = HistState()
    ctx.analysis_reset(96, 80)
    ctx.create_histogram(h["flow"], st)
    assert np.array_equal(st.hist, h["hist"]) and np.array_equal(st.hist2d, h["hist2d"])
    assert st.histsum == h["histsum"].item() and np.array_equal(st.histsum2d, h["histsum2d"])
    assert st.UPPER == h["UPPER"].item() and np.array_equal(st.UPPER2d, h["UPPER2d"])
    assert np.array_equal(st.prop_above_upper, h["prop_above_upper"], equal_nan=True)


@pytest.mark.parametrize("use_graph", [False, True])
def test_lockstep_stream_batch_matches_per_stream(ctx, use_graph):
    """BASELINE config 5 as a parity case: 16 independent streams advanced in lockstep through
    batched launches (hipGraph replay in steady state) give bit-identical flows to 16 separate
    two-image calls."""
    S, T, w, h = 16, 6, 192, 128
    clips = np.stack([synth.surf_clip(w, h, T, seed=100 + s) for s in range(S)])        # [S,T,h,w]
    d = torch.as_tensor(clips).cuda()
    frames = torch.empty((S, h, w), dtype=torch.uint8, device="cuda")                    # fixed staging buffers
    flows = torch.empty((S, h, w, 2), dtype=torch.float32, device="cuda")
    ctx.batch_reset()
    got = []
    for t in range(T):
        frames.copy_(d[:, t])
        r = ctx.push_batch(frames, flows, use_graph=use_graph, **RC215)
        if r is not None:
            ctx.sync()
            got.append(r.cpu().numpy().copy())
    assert len(got) == T - 1
    for s in (0, 7, 15):
        for t in range(T - 1):
            ref = ctx.calcOpticalFlowFarneback(clips[s, t], clips[s, t + 1], None, **RC215)
            assert np.array_equal(got[t][s], ref), (s, t)
    ctx.batch_reset()


def test_config5_sixteen_1080p_streams_hipgraph(ctx, orc):
    """BASELINE config 5 at its stated size: 16 independent 1080p streams advanced in lockstep, hipGraph replay
    in steady state.  Streams 0, 7 and 15 are checked against separate two-image calls (bit-identical) and one of
    them against the oracle (8(d) metric)."""
    S, T, w, h = 16, 5, 1920, 1080
    base = synth.surf_clip(w, h, T + 2, seed=55)
    # 16 different streams from one clip: shifted in time and mirrored / transposed-free variations
    clips = np.stack([np.roll(base[(s % 3):(s % 3) + T], shift=17 * s, axis=2)[:, ::(-1 if s & 1 else 1)].copy() for s in range(S)])
    d = torch.as_tensor(clips).cuda()
    frames = torch.empty((S, h, w), dtype=torch.uint8, device="cuda")
    flows = torch.empty((S, h, w, 2), dtype=torch.float32, device="cuda")
    ctx.batch_reset()
    got = {}
    for t in range(T):
        frames.copy_(d[:, t])
        r = ctx.push_batch(frames, flows, use_graph=True, **RC215)
        if r is not None:
            ctx.sync()
            got[t] = {s: r[s].cpu().numpy().copy() for s in (0, 7, 15)}
    assert sorted(got) == [1, 2, 3, 4]          # eager, capture, replay, replay
    for s in (0, 7, 15):
        for t in (1, 4):
            ref = ctx.calcOpticalFlowFarneback(clips[s, t - 1], clips[s, t], None, **RC215)
            assert np.array_equal(got[t][s], ref), (s, t)
    ref, dets = _oracle_diag(orc, clips[7, 3], clips[7, 4], RC215, nthreads=8)
    assert_conditioned("C5 stream 7 of 16 x 1080p (graph replay)", got[4][7], ref, *dets)
    ctx.batch_reset()


def test_two_threads_on_two_stream_slots(ctx):
    """include/rcflow.h: a context is thread-safe across distinct stream indices.  Two host threads,
    one slot each, different sizes and parameter sets, interleaved calls: same bits as one thread."""
    import threading
    jobs = [(0, (320, 240), RC215, 5), (1, (333, 251), MAIN264, 9)]
    clips = {s: torch.as_tensor(synth.surf_clip(w, h, 6, seed=seed)).cuda() for s, (w, h), _, seed in jobs}
    ref = {}
    for s, (w, h), p, _ in jobs:
        ref[s] = ctx.farneback_clip(clips[s], stream=s, **p).cpu().numpy().copy()
    ctx.sync(0); ctx.sync(1)
    got, errs = {}, []

    def work(s, p):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(8):
                    out = ctx.farneback_clip(clips[s], stream=s, **p)
                st.synchronize()
                got[s] = out.cpu().numpy().copy()
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(s, p)) for s, _, p, _ in jobs]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errs, errs
    for s in ref:
        assert np.array_equal(got[s], ref[s])
