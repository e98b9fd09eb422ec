/*
 * lk_oracle.cpp -- CPU restatement of sparse pyramidal Lucas-Kanade (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * PARITY UNPINNED: cv::calcOpticalFlowPyrLK lives in un-vendored OpenCV 4.1.0
 * (modules/video/src/lkpyramid.cpp: buildOpticalFlowPyramid, calcSharrDeriv, LKTrackerInvoker;
 * imgproc pyrDown), absent from /root/reference and from this image.  This file restates the
 * published algorithm -- the scalar (non-SIMD) path: 8-bit pyramid by pyrDown with REFLECT_101,
 * Scharr derivatives in int16, W_BITS = 14 fixed-point bilinear weights, float accumulators added
 * in raster order.  Reference call sites: Streakline.cpp:32, ripcurrents_module.cpp:775 and :1162
 * (win 50x50, maxLevel 3, COUNT+EPS 30 / 0.1, flags 10, minEig 1e-4); :716, :738 (win 21x21).
 * SURVEY.md section 8(f) row 3.
 */
#include "rc_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
inline int cv_floor(float v) { return (int)std::floor(v); }
inline int cv_round(double v) { return (int)std::nearbyint(v); }     // round half to even
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// One pyramid level: the 8-bit image padded by (bw, bh) with REFLECT_101 and its Scharr
// derivative image (dx, dy interleaved, int16) padded with zeros, like the buffers
// buildOpticalFlowPyramid / calc() hand to LKTrackerInvoker.
struct Level {
    int w = 0, h = 0, bw = 0, bh = 0;
    std::vector<uint8_t> img;      // (h + 2 bh) x (w + 2 bw)
    std::vector<int16_t> der;      // (h + 2 bh) x (w + 2 bw) x 2
    int pitch() const { return w + 2 * bw; }
    const uint8_t* I(int y) const { return img.data() + (size_t)(y + bh) * pitch() + bw; }
    const int16_t* D(int y) const { return der.data() + ((size_t)(y + bh) * pitch() + bw) * 2; }
};

void pad_reflect(const std::vector<uint8_t>& core, int w, int h, Level& L) {
    L.img.resize((size_t)(h + 2 * L.bh) * (w + 2 * L.bw));
    for (int y = -L.bh; y < h + L.bh; y++) {
        const uint8_t* s = core.data() + (size_t)reflect101(y, h) * w;
        uint8_t* d = L.img.data() + (size_t)(y + L.bh) * L.pitch();
        for (int x = -L.bw; x < w + L.bw; x++) d[x + L.bw] = s[reflect101(x, w)];
    }
}

// imgproc pyramids.cpp pyrDown on 8U: (1 4 6 4 1) x (1 4 6 4 1), REFLECT_101, (sum + 128) >> 8
void pyr_down(const std::vector<uint8_t>& src, int sw, int sh, std::vector<uint8_t>& dst, int dw, int dh) {
    dst.resize((size_t)dw * dh);
    std::vector<int> rows((size_t)5 * dw);
    for (int y = 0; y < dh; y++) {
        for (int k = 0; k < 5; k++) {
            const uint8_t* s = src.data() + (size_t)reflect101(2 * y - 2 + k, sh) * sw;
            int* r = rows.data() + (size_t)k * dw;
            for (int x = 0; x < dw; x++) {
                int c = 2 * x;
                r[x] = s[reflect101(c, sw)] * 6 + (s[reflect101(c - 1, sw)] + s[reflect101(c + 1, sw)]) * 4 +
                       s[reflect101(c - 2, sw)] + s[reflect101(c + 2, sw)];
            }
        }
        const int *r0 = rows.data(), *r1 = r0 + dw, *r2 = r1 + dw, *r3 = r2 + dw, *r4 = r3 + dw;
        for (int x = 0; x < dw; x++)
            dst[(size_t)y * dw + x] = (uint8_t)((r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x] + 128) >> 8);
    }
}

// lkpyramid.cpp calcSharrDeriv: dx = d/dx of (3, 10, 3) smoothed columns, dy likewise; REFLECT_101
void scharr(const std::vector<uint8_t>& src, int w, int h, Level& L) {
    L.der.assign((size_t)(h + 2 * L.bh) * L.pitch() * 2, 0);
    std::vector<int> t0(w + 2), t1(w + 2);
    for (int y = 0; y < h; y++) {
        const uint8_t* s0 = src.data() + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * w;
        const uint8_t* s1 = src.data() + (size_t)y * w;
        const uint8_t* s2 = src.data() + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * w;
        int* a = t0.data() + 1;
        int* b = t1.data() + 1;
        for (int x = 0; x < w; x++) {
            a[x] = (s0[x] + s2[x]) * 3 + s1[x] * 10;
            b[x] = s2[x] - s0[x];
        }
        const int x0 = w > 1 ? 1 : 0, x1 = w > 1 ? w - 2 : 0;
        a[-1] = a[x0]; a[w] = a[x1];
        b[-1] = b[x0]; b[w] = b[x1];
        int16_t* d = L.der.data() + ((size_t)(y + L.bh) * L.pitch() + L.bw) * 2;
        for (int x = 0; x < w; x++) {
            d[2 * x] = (int16_t)(a[x + 1] - a[x - 1]);
            d[2 * x + 1] = (int16_t)((b[x + 1] + b[x - 1]) * 3 + b[x] * 10);
        }
    }
}

int build_pyramid(const uint8_t* img, size_t step, int w, int h, int win_w, int win_h, int max_level,
                  bool with_deriv, std::vector<Level>& pyr) {
    std::vector<uint8_t> cur((size_t)w * h), nxt;
    for (int y = 0; y < h; y++) std::memcpy(cur.data() + (size_t)y * w, img + (size_t)y * step, w);
    int cw = w, ch = h;
    pyr.clear();
    for (int level = 0; level <= max_level; level++) {
        Level L;
        L.w = cw; L.h = ch; L.bw = win_w; L.bh = win_h;
        pad_reflect(cur, cw, ch, L);
        if (with_deriv) scharr(cur, cw, ch, L);
        pyr.push_back(std::move(L));
        int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
        if (nw <= win_w || nh <= win_h) return level;      // buildOpticalFlowPyramid's early return
        if (level < max_level) {
            pyr_down(cur, cw, ch, nxt, nw, nh);
            cur.swap(nxt);
            cw = nw; ch = nh;
        }
    }
    return max_level;
}

}  // namespace

extern "C" int orc_pyrdown_u8(const uint8_t* src, size_t step, int w, int h, uint8_t* dst, size_t dst_step) {
    if (!src || !dst || w < 1 || h < 1) return -1;
    std::vector<uint8_t> s((size_t)w * h), d;
    for (int y = 0; y < h; y++) std::memcpy(s.data() + (size_t)y * w, src + (size_t)y * step, w);
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    pyr_down(s, w, h, d, dw, dh);
    for (int y = 0; y < dh; y++) std::memcpy(dst + (size_t)y * dst_step, d.data() + (size_t)y * dw, dw);
    return 0;
}

extern "C" int orc_scharr_deriv(const uint8_t* src, size_t step, int w, int h, int16_t* dxy) {
    if (!src || !dxy || w < 1 || h < 1) return -1;
    std::vector<uint8_t> s((size_t)w * h);
    for (int y = 0; y < h; y++) std::memcpy(s.data() + (size_t)y * w, src + (size_t)y * step, w);
    Level L;
    L.w = w; L.h = h; L.bw = 0; L.bh = 0;
    scharr(s, w, h, L);
    std::memcpy(dxy, L.der.data(), (size_t)w * h * 2 * sizeof(int16_t));
    return 0;
}

extern "C" int orc_pyrlk_levels(int w, int h, int win_w, int win_h, int max_level) {
    int cw = w, ch = h;
    for (int level = 0; level <= max_level; level++) {
        int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
        if (nw <= win_w || nh <= win_h) return level;
        cw = nw; ch = nh;
    }
    return max_level;
}

extern "C" int orc_pyrlk(const uint8_t* prev, size_t prev_step, const uint8_t* next, size_t next_step, int w,
                         int h, const float* prev_pts, float* next_pts, int npts, uint8_t* status, float* err,
                         int win_w, int win_h, int max_level, int crit_type, int max_count, double epsilon,
                         int flags, double min_eig_threshold) {
    if (!prev || !next || !prev_pts || !next_pts || !status || npts < 0 || w < 1 || h < 1 || win_w <= 2 ||
        win_h <= 2 || max_level < 0)
        return -1;
    // SparsePyrLKOpticalFlowImpl::calc: criteria clamping, epsilon squared
    if ((crit_type & 1) == 0) max_count = 30;
    else max_count = std::min(std::max(max_count, 0), 100);
    if ((crit_type & 2) == 0) epsilon = 0.01;
    else epsilon = std::min(std::max(epsilon, 0.), 10.);
    epsilon *= epsilon;
    const bool use_initial = (flags & 4) != 0, get_min_eig = (flags & 8) != 0;

    std::vector<Level> P, N;
    int lp = build_pyramid(prev, prev_step, w, h, win_w, win_h, max_level, true, P);
    int ln = build_pyramid(next, next_step, w, h, win_w, win_h, max_level, false, N);
    max_level = std::min(lp, ln);

    for (int i = 0; i < npts; i++) { status[i] = 1; if (err) err[i] = 0.f; }
    std::vector<int16_t> Ibuf((size_t)win_w * win_h), dIbuf((size_t)win_w * win_h * 2);
    const int W_BITS = 14, W_BITS1 = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float halfx = (win_w - 1) * 0.5f, halfy = (win_h - 1) * 0.5f;

    for (int level = max_level; level >= 0; level--) {
        const Level& I = P[level];
        const Level& J = N[level];
        const int stepI = I.pitch(), stepJ = J.pitch(), dstep = I.pitch() * 2;
        for (int pt = 0; pt < npts; pt++) {
            float px = prev_pts[2 * pt] * (float)(1. / (1 << level));
            float py = prev_pts[2 * pt + 1] * (float)(1. / (1 << level));
            float nx, ny;
            if (level == max_level) {
                if (use_initial) {
                    nx = next_pts[2 * pt] * (float)(1. / (1 << level));
                    ny = next_pts[2 * pt + 1] * (float)(1. / (1 << level));
                } else { nx = px; ny = py; }
            } else {
                nx = next_pts[2 * pt] * 2.f;
                ny = next_pts[2 * pt + 1] * 2.f;
            }
            next_pts[2 * pt] = nx; next_pts[2 * pt + 1] = ny;

            px -= halfx; py -= halfy;
            int ipx = cv_floor(px), ipy = cv_floor(py);
            if (ipx < -win_w || ipx >= I.w || ipy < -win_h || ipy >= I.h) {
                if (level == 0) { status[pt] = 0; if (err) err[pt] = 0.f; }
                continue;
            }
            float a = px - ipx, b = py - ipy;
            int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            float iA11 = 0, iA12 = 0, iA22 = 0;
            for (int y = 0; y < win_h; y++) {
                const uint8_t* src = I.I(y + ipy) + ipx;
                const int16_t* dsrc = I.D(y + ipy) + ipx * 2;
                int16_t* Ip = Ibuf.data() + (size_t)y * win_w;
                int16_t* dIp = dIbuf.data() + (size_t)y * win_w * 2;
                for (int x = 0; x < win_w; x++, dsrc += 2, dIp += 2) {
                    int ival = descale(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 +
                                       src[x + stepI + 1] * iw11, W_BITS1 - 5);
                    int ixval = descale(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 + dsrc[dstep + 2] * iw11,
                                        W_BITS1);
                    int iyval = descale(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[dstep + 1] * iw10 +
                                        dsrc[dstep + 3] * iw11, W_BITS1);
                    Ip[x] = (int16_t)ival;
                    dIp[0] = (int16_t)ixval;
                    dIp[1] = (int16_t)iyval;
                    iA11 += (float)(ixval * ixval);
                    iA12 += (float)(ixval * iyval);
                    iA22 += (float)(iyval * iyval);
                }
            }
            float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                           (2 * win_w * win_h);
            if (err && get_min_eig) err[pt] = minEig;
            if (minEig < min_eig_threshold || D < FLT_EPSILON) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= halfx; ny -= halfy;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_count; j++) {
                int inx = cv_floor(nx), iny = cv_floor(ny);
                if (inx < -win_w || inx >= J.w || iny < -win_h || iny >= J.h) {
                    if (level == 0) status[pt] = 0;
                    break;
                }
                a = nx - inx; b = ny - iny;
                iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                float ib1 = 0, ib2 = 0;
                for (int y = 0; y < win_h; y++) {
                    const uint8_t* Jp = J.I(y + iny) + inx;
                    const int16_t* Ip = Ibuf.data() + (size_t)y * win_w;
                    const int16_t* dIp = dIbuf.data() + (size_t)y * win_w * 2;
                    for (int x = 0; x < win_w; x++, dIp += 2) {
                        int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 +
                                           Jp[x + stepJ + 1] * iw11, W_BITS1 - 5) - Ip[x];
                        ib1 += (float)(diff * dIp[0]);
                        ib2 += (float)(diff * dIp[1]);
                    }
                }
                float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
                float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
                nx += dx; ny += dy;
                next_pts[2 * pt] = nx + halfx; next_pts[2 * pt + 1] = ny + halfy;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;      // delta.ddot(delta)
                if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01) {
                    next_pts[2 * pt] -= dx * 0.5f;
                    next_pts[2 * pt + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[pt] && err && level == 0 && !get_min_eig) {
                float fx = next_pts[2 * pt] - halfx, fy = next_pts[2 * pt + 1] - halfy;
                int inx = cv_floor(fx), iny = cv_floor(fy);
                if (inx < -win_w || inx >= J.w || iny < -win_h || iny >= J.h) {
                    status[pt] = 0;
                    continue;
                }
                float aa = fx - inx, bb = fy - iny;
                iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
                iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
                iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                float errval = 0.f;
                for (int y = 0; y < win_h; y++) {
                    const uint8_t* Jp = J.I(y + iny) + inx;
                    const int16_t* Ip = Ibuf.data() + (size_t)y * win_w;
                    for (int x = 0; x < win_w; x++) {
                        int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 +
                                           Jp[x + stepJ + 1] * iw11, W_BITS1 - 5) - Ip[x];
                        errval += std::abs((float)diff);
                    }
                }
                err[pt] = errval * 1.f / (32 * win_w * win_h);
            }
        }
    }
    return 0;
}

/* Streakline.cpp:22-71 with the reference's own mover: vertices advanced by PyrLK
 * (win 50x50, maxLevel 3, 30 iterations / eps 0.1, flags 10, minEig 1e-4), jumps above a tenth
 * of the frame reverted, generation point inserted at index 0, frame counter advanced. */
extern "C" int orc_streakline_step_lk(float* verts, int* nverts, float gen_x, float gen_y, const uint8_t* prev,
                                      size_t prev_step, const uint8_t* next, size_t next_step, int w, int h,
                                      int* frame_count) {
    int n = *nverts;
    std::vector<float> nextv((size_t)2 * std::max(n, 1));
    std::vector<uint8_t> st(std::max(n, 1));
    std::vector<float> er(std::max(n, 1));
    if (n > 0) {
        int rc = orc_pyrlk(prev, prev_step, next, next_step, w, h, verts, nextv.data(), n, st.data(), er.data(), 50,
                           50, 3, 3, 30, 0.1, 10, 1e-4);
        if (rc) return rc;
        for (int i = 0; i < n; i++) {
            if (std::abs(verts[2 * i] - nextv[2 * i]) > w * 0.1 || std::abs(verts[2 * i + 1] - nextv[2 * i + 1]) > h * 0.1) {
                nextv[2 * i] = verts[2 * i];
                nextv[2 * i + 1] = verts[2 * i + 1];
            }
        }
    }
    // vertices = vertices_next; insert the generation point at the front (Streakline.cpp:43-48)
    for (int i = n - 1; i >= 0; i--) { verts[2 * (i + 1)] = nextv[2 * i]; verts[2 * (i + 1) + 1] = nextv[2 * i + 1]; }
    verts[0] = gen_x; verts[1] = gen_y;
    *nverts = n + 1;
    (*frame_count)++;
    return 0;
}
