"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT.  PARITY UNPINNED: the reference has no golden
vectors for this path and its Farneback arithmetic lives in un-vendored OpenCV
4.1.0; see oracle/rc_oracle.h.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FARNEBACK_GAUSSIAN = 256
HIST_BINS, HIST_DIRECTIONS, HIST_RESOLUTION = 50, 36, 20


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_farneback_u8.restype = C.c_int
        _LIB.orc_level_geometry.restype = C.c_int
    return _LIB


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def farneback(prev, nxt, pyr_scale=0.5, levels=2, winsize=3, iters=2, poly_n=15,
              poly_sigma=1.2, flags=0, nthreads=1):
    """cv::calcOpticalFlowFarneback on two HxW uint8 images -> HxWx2 float32."""
    prev = np.ascontiguousarray(prev, dtype=np.uint8)
    nxt = np.ascontiguousarray(nxt, dtype=np.uint8)
    h, w = prev.shape
    flow = np.zeros((h, w, 2), np.float32)
    rc = lib().orc_farneback_u8(
        _p(prev, C.c_uint8), C.c_size_t(prev.strides[0]), _p(nxt, C.c_uint8),
        C.c_size_t(nxt.strides[0]), w, h, _p(flow), C.c_size_t(flow.strides[0]),
        C.c_double(pyr_scale), levels, winsize, iters, poly_n, C.c_double(poly_sigma), flags,
        nthreads)
    if rc != 0:
        raise ValueError("orc_farneback_u8 rejected its arguments (rc=%d)" % rc)
    return flow


class _Diag(C.Structure):
    _fields_ = [("g_last", C.POINTER(C.c_float)), ("det_min", C.POINTER(C.c_float)),
                ("level_flow", C.POINTER(C.c_float) * 12)]


def farneback_diag(prev, nxt, pyr_scale=0.5, levels=2, winsize=3, iters=2, poly_n=15,
                   poly_sigma=1.2, flags=0, nthreads=1, level_flows=False):
    """farneback() plus the conditioning diagnostics of orc_farneback_u8_ex:
    returns (flow, det_last, det_min[, level_flows]) with det_last = g11*g22 - g12^2 of the final
    solve at every pixel (float64) and det_min its minimum over the pixel's whole coarse-to-fine path."""
    prev = np.ascontiguousarray(prev, dtype=np.uint8)
    nxt = np.ascontiguousarray(nxt, dtype=np.uint8)
    h, w = prev.shape
    flow = np.zeros((h, w, 2), np.float32)
    g_last = np.zeros((h, w, 3), np.float32)
    det_min = np.zeros((h, w), np.float32)
    d = _Diag()
    d.g_last = _p(g_last)
    d.det_min = _p(det_min)
    lf = []
    if level_flows:
        L = level_geometry(w, h, pyr_scale, levels, 0)["levels"]
        for k in range(L + 1):
            g = level_geometry(w, h, pyr_scale, levels, k)
            lf.append(np.zeros((g["h"], g["w"], 2), np.float32))
            d.level_flow[k] = _p(lf[k])
    fn = lib().orc_farneback_u8_ex
    fn.restype = C.c_int
    rc = fn(_p(prev, C.c_uint8), C.c_size_t(prev.strides[0]), _p(nxt, C.c_uint8),
            C.c_size_t(nxt.strides[0]), w, h, _p(flow), C.c_size_t(flow.strides[0]),
            C.c_double(pyr_scale), levels, winsize, iters, poly_n, C.c_double(poly_sigma), flags,
            nthreads, C.byref(d))
    if rc != 0:
        raise ValueError("orc_farneback_u8_ex rejected its arguments (rc=%d)" % rc)
    g = g_last.astype(np.float64)
    det_last = g[..., 0] * g[..., 2] - g[..., 1] ** 2
    return (flow, det_last, det_min, lf) if level_flows else (flow, det_last, det_min)


def level_geometry(w, h, pyr_scale, levels, k):
    wk, hk, ks = C.c_int(), C.c_int(), C.c_int()
    sg = C.c_double()
    L = lib().orc_level_geometry(w, h, C.c_double(pyr_scale), levels, k, C.byref(wk),
                                 C.byref(hk), C.byref(sg), C.byref(ks))
    return dict(levels=L, w=wk.value, h=hk.value, sigma=sg.value, ksize=ks.value)


def gaussian_kernel(n, sigma):
    k = np.zeros(n, np.float32)
    lib().orc_gaussian_kernel(n, C.c_double(sigma), _p(k))
    return k


def gaussian_blur(src, ksize, sigma):
    src = _f32(src)
    h, w = src.shape
    dst = np.empty_like(src)
    lib().orc_gaussian_blur_f32(_p(src), w, h, ksize, C.c_double(sigma), _p(dst))
    return dst


def resize_linear(src, dw, dh):
    src = _f32(src)
    cn = 1 if src.ndim == 2 else src.shape[2]
    sh, sw = src.shape[:2]
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.float32)
    lib().orc_resize_linear_f32(_p(src), sw, sh, cn, _p(dst), dw, dh)
    return dst


def pyr_level(img, sigma, ksize, ow, oh):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((oh, ow), np.float32)
    lib().orc_pyr_level(_p(img, C.c_uint8), C.c_size_t(img.strides[0]), w, h, C.c_double(sigma),
                        ksize, _p(out), ow, oh)
    return out


def prepare_gaussian(n, sigma):
    g = np.zeros(2 * n + 1, np.float32)
    xg = np.zeros_like(g)
    xxg = np.zeros_like(g)
    ig = np.zeros(4, np.float64)
    lib().orc_prepare_gaussian(n, C.c_double(sigma), _p(g), _p(xg), _p(xxg), _p(ig, C.c_double))
    return g, xg, xxg, ig


def polyexp(I, n=15, sigma=1.2):
    I = _f32(I)
    h, w = I.shape
    R = np.empty((h, w, 5), np.float32)
    lib().orc_polyexp(_p(I), w, h, n, C.c_double(sigma), _p(R))
    return R


def update_matrices(R0, R1, flow, M=None, y0=0, y1=None):
    R0, R1, flow = _f32(R0), _f32(R1), _f32(flow)
    h, w = flow.shape[:2]
    if M is None:
        M = np.zeros((h, w, 5), np.float32)
    lib().orc_update_matrices(_p(R0), _p(R1), _p(flow), _p(M), w, h, y0, h if y1 is None else y1)
    return M


def update_flow(R0, R1, flow, M, block_size, update_matrices_flag, gaussian):
    """In-place on flow and M (both float32 C-contiguous)."""
    h, w = flow.shape[:2]
    fn = lib().orc_update_flow_gaussian if gaussian else lib().orc_update_flow_blur
    fn(_p(R0), _p(R1), _p(flow), _p(M), w, h, block_size, int(update_matrices_flag))
    return flow, M


# ------------------------------------------------------------------ B rows
def fast_atan2_deg(y, x):
    y, x = _f32(y).ravel(), _f32(x).ravel()
    out = np.empty_like(x)
    lib().orc_fast_atan2_deg(_p(y), _p(x), _p(out), x.size)
    return out


def flow_to_polar(flow):
    flow = _f32(flow)
    h, w = flow.shape[:2]
    polar = np.empty((h, w, 3), np.float32)
    lib().orc_flow_to_polar(_p(flow), C.c_size_t(flow.strides[0]), w, h, _p(polar),
                            C.c_size_t(polar.strides[0]))
    return polar


class HistState:
    """The caller-owned arrays of create_histogram (ripcurrents.cpp:147-154)."""

    def __init__(self):
        self.hist = np.zeros(HIST_BINS, np.int32)
        self.histsum = C.c_int32(0)
        self.hist2d = np.zeros((HIST_DIRECTIONS, HIST_BINS), np.int32)
        self.histsum2d = np.zeros(HIST_DIRECTIONS, np.int32)
        self.UPPER = 100.0
        self.UPPER2d = np.zeros(HIST_DIRECTIONS, np.float32)
        self.prop_above_upper = np.zeros(HIST_DIRECTIONS, np.float32)


def histogram_accumulate(polar, st):
    polar = _f32(polar)
    h, w = polar.shape[:2]
    lib().orc_histogram_accumulate(_p(polar), C.c_size_t(polar.strides[0]), w, h,
                                   _p(st.hist, C.c_int32), C.byref(st.histsum),
                                   _p(st.hist2d, C.c_int32), _p(st.histsum2d, C.c_int32))


def histogram_thresholds(st):
    up = C.c_float()
    lib().orc_histogram_thresholds(_p(st.hist, C.c_int32), st.histsum, _p(st.hist2d, C.c_int32),
                                   _p(st.histsum2d, C.c_int32), C.byref(up), _p(st.UPPER2d),
                                   _p(st.prop_above_upper))
    st.UPPER = up.value


def create_histogram(polar, st):
    """ripcurrents_module.cpp:89-144"""
    histogram_accumulate(polar, st)
    histogram_thresholds(st)


def create_flow(polar, waterclass, accumulator2, UPPER, MID, LOWER, UPPER2d):
    h, w = polar.shape[:2]
    lib().orc_create_flow(_p(polar), C.c_size_t(polar.strides[0]), _p(waterclass),
                          C.c_size_t(waterclass.strides[0]), _p(accumulator2),
                          C.c_size_t(accumulator2.strides[0]), w, h, C.c_float(UPPER),
                          C.c_float(MID), C.c_float(LOWER), _p(_f32(UPPER2d)))


def create_accumulationbuffer(accumulator, accumulator2, out, outmask, framecount):
    h, w = accumulator.shape[:2]
    lib().orc_create_accumulationbuffer(
        _p(accumulator), C.c_size_t(accumulator.strides[0]), _p(accumulator2),
        C.c_size_t(accumulator2.strides[0]), _p(out), C.c_size_t(out.strides[0]),
        _p(outmask, C.c_uint8), C.c_size_t(outmask.strides[0]), w, h, framecount)


def streamline_field(pt, dist, flow, dt, iterations, UPPER):
    h, w = flow.shape[:2]
    lib().orc_streamline_field(_p(pt), C.c_size_t(pt.strides[0]), _p(dist),
                               C.c_size_t(dist.strides[0]), _p(flow),
                               C.c_size_t(flow.strides[0]), w, h, C.c_float(dt), iterations,
                               C.c_float(UPPER))


def streamline_points(pts, flow, dt, iterations, UPPER, variant=0, trace=False):
    h, w = flow.shape[:2]
    n = pts.shape[0]
    iters = 100 if variant == 2 else iterations
    tr = np.zeros((n, iters, 2), np.float32) if trace else None
    lib().orc_streamline_points(_p(pts), n, _p(flow), C.c_size_t(flow.strides[0]), w, h,
                                C.c_float(dt), iterations, C.c_float(UPPER), variant,
                                _p(tr) if trace else None)
    return tr


def get_delta_field(pt, flow, dt, UPPER):
    h, w = flow.shape[:2]
    lib().orc_get_delta_field(_p(pt), C.c_size_t(pt.strides[0]), _p(flow),
                              C.c_size_t(flow.strides[0]), w, h, C.c_float(dt), C.c_float(UPPER))


def streakline_step(verts, nverts, gen, flow, dt, frame_count):
    h, w = flow.shape[:2]
    n = C.c_int(nverts)
    fc = C.c_int(frame_count)
    lib().orc_streakline_step(_p(verts), C.byref(n), C.c_float(gen[0]), C.c_float(gen[1]),
                              _p(flow), C.c_size_t(flow.strides[0]), w, h, C.c_float(dt),
                              C.byref(fc))
    return n.value, fc.value


def subtract_average(flow):
    h, w = flow.shape[:2]
    lib().orc_subtract_average(_p(flow), C.c_size_t(flow.strides[0]), w, h)


def subtract_mean_magnitude(flow):
    h, w = flow.shape[:2]
    lib().orc_subtract_mean_magnitude(_p(flow), C.c_size_t(flow.strides[0]), w, h)


def stabilizer(flow):
    h, w = flow.shape[:2]
    lib().orc_stabilizer(_p(flow), C.c_size_t(flow.strides[0]), w, h)


def window_mean_update(avg, slot, cur, window):
    lib().orc_window_mean_update(_p(avg), _p(slot), _p(_f32(cur)), avg.size, window)


def vector_to_color(flow, max_displacement):
    h, w = flow.shape[:2]
    hsv = np.zeros((h, w, 3), np.uint8)
    md = C.c_float(max_displacement)
    lib().orc_vector_to_color(_p(flow), C.c_size_t(flow.strides[0]), w, h, _p(hsv, C.c_uint8),
                              C.c_size_t(hsv.strides[0]), C.byref(md))
    return hsv, md.value


def shear_rate_to_color(flow, max_frobenius, hsv=None):
    h, w = flow.shape[:2]
    if hsv is None:
        hsv = np.zeros((h, w, 3), np.uint8)
    mf = C.c_float(max_frobenius)
    lib().orc_shear_rate_to_color(_p(flow), C.c_size_t(flow.strides[0]), w, h,
                                  _p(hsv, C.c_uint8), C.c_size_t(hsv.strides[0]), C.byref(mf))
    return hsv, mf.value


def create_edges(mask):
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    out = np.zeros_like(mask)
    lib().orc_create_edges(_p(mask, C.c_uint8), C.c_size_t(mask.strides[0]), w, h, _p(out, C.c_uint8),
                           C.c_size_t(out.strides[0]))
    return out


def ellipse5():
    k = np.zeros(25, np.uint8)
    lib().orc_ellipse5(_p(k, C.c_uint8))
    return k.reshape(5, 5)


def resize_bgr_to_gray(bgr, dw, dh):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    sh, sw = bgr.shape[:2]
    gray = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_bgr_to_gray(_p(bgr, C.c_uint8), C.c_size_t(bgr.strides[0]), sw, sh, _p(gray, C.c_uint8),
                                 C.c_size_t(gray.strides[0]), dw, dh)
    return gray


# ---- section 8(f) row 3: sparse pyramidal Lucas-Kanade (lk_oracle.cpp)
LK_USE_INITIAL_FLOW, LK_GET_MIN_EIGENVALS = 4, 8
CRIT_COUNT, CRIT_EPS = 1, 2


def pyrlk(prev, nxt, prev_pts, next_pts=None, win=(21, 21), max_level=3, crit_type=CRIT_COUNT | CRIT_EPS,
          max_count=30, epsilon=0.01, flags=0, min_eig_threshold=1e-4):
    """cv::calcOpticalFlowPyrLK on two HxW uint8 images.  Returns (next_pts, status, err)."""
    prev = np.ascontiguousarray(prev, dtype=np.uint8)
    nxt = np.ascontiguousarray(nxt, dtype=np.uint8)
    h, w = prev.shape
    p = _f32(prev_pts).reshape(-1, 2)
    n = p.shape[0]
    q = np.zeros((n, 2), np.float32) if next_pts is None else _f32(next_pts).reshape(-1, 2).copy()
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    rc = lib().orc_pyrlk(_p(prev, C.c_uint8), C.c_size_t(prev.strides[0]), _p(nxt, C.c_uint8),
                         C.c_size_t(nxt.strides[0]), w, h, _p(p), _p(q), n, _p(status, C.c_uint8), _p(err),
                         win[0], win[1], max_level, crit_type, max_count, C.c_double(epsilon), flags,
                         C.c_double(min_eig_threshold))
    if rc != 0:
        raise ValueError("orc_pyrlk rejected its arguments (rc=%d)" % rc)
    return q, status, err


def pyrlk_levels(w, h, win, max_level):
    return lib().orc_pyrlk_levels(w, h, win[0], win[1], max_level)


def pyrdown_u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyrdown_u8(_p(img, C.c_uint8), C.c_size_t(img.strides[0]), w, h, _p(out, C.c_uint8),
                         C.c_size_t(out.strides[0]))
    return out


def scharr_deriv(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros((h, w, 2), np.int16)
    lib().orc_scharr_deriv(_p(img, C.c_uint8), C.c_size_t(img.strides[0]), w, h, _p(out, C.c_int16))
    return out


def streakline_step_lk(verts, nverts, gen, prev, nxt, frame_count):
    prev = np.ascontiguousarray(prev, dtype=np.uint8)
    nxt = np.ascontiguousarray(nxt, dtype=np.uint8)
    h, w = prev.shape
    n = C.c_int(nverts)
    fc = C.c_int(frame_count)
    rc = lib().orc_streakline_step_lk(_p(verts), C.byref(n), C.c_float(gen[0]), C.c_float(gen[1]),
                                      _p(prev, C.c_uint8), C.c_size_t(prev.strides[0]), _p(nxt, C.c_uint8),
                                      C.c_size_t(nxt.strides[0]), w, h, C.byref(fc))
    if rc != 0:
        raise ValueError("orc_streakline_step_lk failed (rc=%d)" % rc)
    return n.value, fc.value


# ---- section 8(f) row 4: display path (display_oracle.cpp)
def jet_lut():
    lut = np.zeros((256, 3), np.uint8)
    lib().orc_jet_lut(_p(lut, C.c_uint8))
    return lut


def streamline_display(pt, dist, which):
    pt, dist = _f32(pt), _f32(dist)
    h, w = dist.shape
    bgr = np.zeros((h, w, 3), np.uint8)
    mx = C.c_double(0)
    lib().orc_streamline_display(_p(pt), C.c_size_t(pt.strides[0]), _p(dist), C.c_size_t(dist.strides[0]), w, h,
                                 which, _p(bgr, C.c_uint8), C.c_size_t(bgr.strides[0]), C.byref(mx))
    return bgr, mx.value


def streamline_positions(pt):
    pt = _f32(pt)
    h, w = pt.shape[:2]
    den = np.zeros((h, w, 3), np.float32)
    lib().orc_streamline_positions(_p(pt), C.c_size_t(pt.strides[0]), w, h, _p(den), C.c_size_t(den.strides[0]))
    return den


def hsv_to_bgr(hsv):
    hsv = _f32(hsv)
    h, w = hsv.shape[:2]
    out = np.zeros((h, w, 3), np.float32)
    lib().orc_hsv_to_bgr_f32(_p(hsv), C.c_size_t(hsv.strides[0]), w, h, _p(out), C.c_size_t(out.strides[0]))
    return out


def resize_area_bgr_to_gray(bgr, dw, dh):
    """resize(frame, Size(dw, dh), INTER_AREA) + cvtColor(BGR2GRAY) (the first frame, ripcurrents.cpp:186)."""
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    sh, sw = bgr.shape[:2]
    out = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_area_bgr_to_gray(_p(bgr, C.c_uint8), C.c_size_t(bgr.strides[0]), sw, sh, _p(out, C.c_uint8),
                                      C.c_size_t(out.strides[0]), dw, dh)
    return out
