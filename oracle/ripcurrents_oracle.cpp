// ripcurrents_oracle.cpp -- CPU restatement of the per-pixel analysis downstream of the
// flow field (B1-B8).  TEST INFRASTRUCTURE, NOT PRODUCT.  PARITY UNPINNED (rc_oracle.h).
//
// Follows, function by function (paths relative to /root/reference/RipCurrents_main):
//   B1 ripcurrents.cpp:305-309          split / cartToPolar(deg) / merge
//      (cartToPolar is OpenCV core 4.1.0: hal::magnitude32f + hal::fastAtan32f)
//   B2 ripcurrents_module.cpp:89-144    create_histogram
//   B3 ripcurrents_module.cpp:153-212   create_flow, create_accumulationbuffer
//   B4 ripcurrents_module.cpp:608-648   streamline_field
//   B5 ripcurrents_module.cpp:486-606,650-679; ripcurrents.cpp:656-698; pathlines.cpp:9-46
//   B6 Streakline.cpp:11-71             Streakline bookkeeping
//   B7 ripcurrents_module.cpp:279-308,810-1015; main.cpp:1142-1153
//   B8 ripcurrents_module.cpp:1017-1138 vectorToColor, shearRateToColor
// Built with -ffp-contract=off: every float expression rounds as written.

#include "rc_oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

// OpenCV core mathfuncs_core: atan_f32, degrees, ~0.3 deg accurate polynomial.
const float atan2_p1 = 0.9997878412794807f * (float)(180 / M_PI);
const float atan2_p3 = -0.3258083974640975f * (float)(180 / M_PI);
const float atan2_p5 = 0.1555786518463281f * (float)(180 / M_PI);
const float atan2_p7 = -0.04432655554792128f * (float)(180 / M_PI);

inline float fast_atan2_deg(float y, float x) {
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

template <typename T>
inline T* row_ptr(T* base, size_t step, int y) {
    return (T*)((char*)base + (size_t)y * step);
}
template <typename T>
inline const T* row_ptr(const T* base, size_t step, int y) {
    return (const T*)((const char*)base + (size_t)y * step);
}

// direction index of ripcurrents_module.cpp:100,158: int angle = (a*36)/360
inline int dir_index(float angle) {
    int d = (int)((angle * ORC_HIST_DIRECTIONS) / 360);
    // angle==360.0f gives 36: out of bounds in the reference; folded to 0 here.
    if (d >= ORC_HIST_DIRECTIONS) d = 0;
    if (d < 0) d = 0;
    return d;
}

// x86 float -> uchar store as compiled from `uchar = float_expr` (cvttss2si, low byte)
inline uint8_t f2u8(float v) {
    int iv;
    if (!(v > -2147483904.f && v < 2147483648.f)) iv = INT_MIN;  // NaN / overflow
    else iv = (int)v;
    return (uint8_t)(iv & 0xFF);
}

// Shared bilinear sampler of ripcurrents_module.cpp:494-508 (and every copy of it).
// Returns false when the bounds check rejects the position.
inline bool sample_flow(const float* flow, size_t step, int w, int h, float x, float y,
                        float& dx, float& dy) {
    int xind = (int)floorf(x);
    int yind = (int)floorf(y);
    float xrem = x - xind;
    float yrem = y - yind;
    if (xind < 1 || yind < 1 || xind + 2 > w || yind + 2 > h) return false;
    const float* r0 = row_ptr(flow, step, yind) + 2 * xind;
    const float* r1 = row_ptr(flow, step, yind + 1) + 2 * xind;
    // Pixel2 * float, left-to-right: ((p*(1-xrem))*(1-yrem)) + ...
    float w00a = (1 - xrem), w00b = (1 - yrem);
    dx = r0[0] * w00a * w00b + r0[2] * xrem * w00b + r1[0] * w00a * yrem + r1[2] * xrem * yrem;
    dy = r0[1] * w00a * w00b + r0[3] * xrem * w00b + r1[1] * w00a * yrem + r1[3] * xrem * yrem;
    return true;
}

}  // namespace

extern "C" {

void orc_fast_atan2_deg(const float* y, const float* x, float* angle, int n) {
    for (int i = 0; i < n; i++) angle[i] = fast_atan2_deg(y[i], x[i]);
}

// B1: current = merge(angle_deg, mag, mag)
void orc_flow_to_polar(const float* flow, size_t flow_step, int w, int h, float* polar,
                       size_t polar_step) {
    for (int y = 0; y < h; y++) {
        const float* f = row_ptr(flow, flow_step, y);
        float* p = row_ptr(polar, polar_step, y);
        for (int x = 0; x < w; x++) {
            float fx = f[2 * x], fy = f[2 * x + 1];
            float mag = std::sqrt(fx * fx + fy * fy);
            p[3 * x] = fast_atan2_deg(fy, fx);
            p[3 * x + 1] = mag;
            p[3 * x + 2] = mag;
        }
    }
}

// B2 counts: ripcurrents_module.cpp:94-107
void orc_histogram_accumulate(const float* polar, size_t polar_step, int w, int h,
                              int32_t* hist, int32_t* histsum, int32_t* hist2d,
                              int32_t* histsum2d) {
    for (int y = 0; y < h; y++) {
        const float* p = row_ptr(polar, polar_step, y);
        for (int x = 0; x < w; x++) {
            int bin = (int)(p[3 * x + 1] * ORC_HIST_RESOLUTION);
            int angle = dir_index(p[3 * x]);
            if (bin < ORC_HIST_BINS && bin >= 0) {
                hist[bin]++;
                (*histsum)++;
                hist2d[angle * ORC_HIST_BINS + bin]++;
                histsum2d[angle]++;
            }
        }
    }
}

// B2 thresholds: ripcurrents_module.cpp:109-144
void orc_histogram_thresholds(const int32_t* hist, int32_t histsum, const int32_t* hist2d,
                              const int32_t* histsum2d, float* UPPER, float* UPPER2d,
                              float* prop_above_upper) {
    int threshsum = 0;
    int bin = ORC_HIST_BINS - 1;
    while (threshsum < (histsum * .05)) {
        threshsum += hist[bin];
        bin--;
    }
    *UPPER = bin / float(ORC_HIST_RESOLUTION);
    int targetbin = bin;
    for (int angle = 0; angle < ORC_HIST_DIRECTIONS; angle++) {
        int threshsum2 = 0;
        int b = ORC_HIST_BINS - 1;
        while (threshsum2 < (histsum2d[angle] * .05)) {
            threshsum2 += hist2d[angle * ORC_HIST_BINS + b];
            b--;
        }
        UPPER2d[angle] = b / float(ORC_HIST_RESOLUTION);
        if (UPPER2d[angle] < 0.01) UPPER2d[angle] = 0.01;
        int threshsum3 = 0;
        b = ORC_HIST_BINS - 1;
        while (b > targetbin) {
            threshsum3 += hist2d[angle * ORC_HIST_BINS + b];
            b--;
        }
        prop_above_upper[angle] = ((float)threshsum3) / threshsum;  // 0/0 -> NaN, as there
    }
}

// B3a: ripcurrents_module.cpp:153-182
void orc_create_flow(float* polar, size_t polar_step, float* waterclass, size_t wc_step,
                     float* accumulator2, size_t acc2_step, int w, int h, float UPPER,
                     float MID, float LOWER, const float* UPPER2d) {
    for (int y = 0; y < h; y++) {
        float* p = row_ptr(polar, polar_step, y);
        float* wc = row_ptr(waterclass, wc_step, y);
        float* a2 = row_ptr(accumulator2, acc2_step, y);
        for (int x = 0; x < w; x++) {
            int angle = dir_index(p[3 * x]);
            float val = p[3 * x + 2];
            if (val > UPPER) { wc[3 * x] = .5; a2[3 * x]++; }
            else if (val > MID) { wc[3 * x + 2] = 1; }
            else if (val > LOWER) { wc[3 * x + 2] = .5; }
            else { wc[3 * x + 1] = .5; }
            p[3 * x + 2] = val / UPPER2d[angle];
            if (p[3 * x + 2] > 1) p[3 * x + 1] = 1;
            else p[3 * x + 1] = .7;
        }
    }
}

// B3b: ripcurrents_module.cpp:189-212
void orc_create_accumulationbuffer(float* accumulator, size_t acc_step,
                                   const float* accumulator2, size_t acc2_step, float* out,
                                   size_t out_step, uint8_t* outmask, size_t mask_step,
                                   int w, int h, int framecount) {
    for (int y = 0; y < h; y++) {
        float* acc = row_ptr(accumulator, acc_step, y);
        const float* a2 = row_ptr(accumulator2, acc2_step, y);
        float* o = row_ptr(out, out_step, y);
        uint8_t* mk = row_ptr(outmask, mask_step, y);
        for (int x = 0; x < w; x++) {
            if (framecount > 30)
                for (int c = 0; c < 3; c++) acc[3 * x + c] = a2[3 * x + c] + acc[3 * x + c];
            int val = (int)acc[3 * x];
            if (val > .1 * framecount) {
                if (val < .2 * framecount) o[3 * x + 2] = 1;
                else o[3 * x] = 1;
            } else {
                o[3 * x + 1] = .5;
                mk[x] = 255;
            }
        }
    }
}

// B4: ripcurrents_module.cpp:608-648 for every pixel (call: ripcurrents.cpp:229-231)
void orc_streamline_field(float* pt, size_t pt_step, float* dist, size_t dist_step,
                          const float* flow, size_t flow_step, int w, int h, float dt,
                          int iterations, float UPPER) {
    for (int yo = 0; yo < h; yo++) {
        float* p = row_ptr(pt, pt_step, yo);
        float* d = row_ptr(dist, dist_step, yo);
        for (int xo = 0; xo < w; xo++) {
            for (int i = 0; i < iterations; i++) {
                float x = p[2 * xo] + xo;
                float y = p[2 * xo + 1] + yo;
                float dx, dy;
                if (!sample_flow(flow, flow_step, w, h, x, y, dx, dy)) break;
                float r = std::sqrt(dx * dx + dy * dy);
                if (r > UPPER) break;
                // *pt + delta*dt/iterations
                p[2 * xo] = p[2 * xo] + dx * dt / iterations;
                p[2 * xo + 1] = p[2 * xo + 1] + dy * dt / iterations;
                d[xo] = d[xo] + r;
            }
        }
    }
}

// B5: seed lists (line drawing stays on the host and is not restated)
void orc_streamline_points(float* pts, int n, const float* flow, size_t flow_step, int w,
                           int h, float dt, int iterations, float UPPER, int variant,
                           float* trace) {
    int iters = variant == 2 ? 100 : iterations;
    for (int s = 0; s < n; s++) {
        bool alive = true;
        for (int i = 0; i < iters; i++) {
            float x = pts[2 * s], y = pts[2 * s + 1];
            float dx, dy;
            if (alive && !sample_flow(flow, flow_step, w, h, x, y, dx, dy)) alive = false;
            if (alive) {
                float r = std::sqrt(dx * dx + dy * dy);
                if ((variant == 0 || variant == 3) && r > UPPER) alive = false;
                if (variant == 1 && r > 5) alive = false;
            }
            if (alive) {
                switch (variant) {
                    case 0: case 1:
                        pts[2 * s] = x + dx * dt;
                        pts[2 * s + 1] = y + dy * dt;
                        break;
                    case 2:
                        // delta*0.1: Point_<float> * double -> saturate_cast<float>(v*0.1)
                        pts[2 * s] = x + (float)(dx * 0.1);
                        pts[2 * s + 1] = y + (float)(dy * 0.1);
                        break;
                    default:
                        pts[2 * s] = x + dx * dt / iterations;
                        pts[2 * s + 1] = y + dy * dt / iterations;
                }
            }
            if (trace) {
                trace[((size_t)s * iters + i) * 2] = pts[2 * s];
                trace[((size_t)s * iters + i) * 2 + 1] = pts[2 * s + 1];
            }
        }
    }
}

// get_delta: ripcurrents_module.cpp:650-679 for every pixel
void orc_get_delta_field(float* pt, size_t pt_step, const float* flow, size_t flow_step,
                         int w, int h, float dt, float UPPER) {
    for (int yo = 0; yo < h; yo++) {
        float* p = row_ptr(pt, pt_step, yo);
        for (int xo = 0; xo < w; xo++) {
            float x = p[2 * xo] + xo, y = p[2 * xo + 1] + yo;
            float dx, dy;
            if (!sample_flow(flow, flow_step, w, h, x, y, dx, dy)) continue;
            float r = std::sqrt(dx * dx + dy * dy);
            if (r > UPPER) continue;
            p[2 * xo] = p[2 * xo] + dx * dt;
            p[2 * xo + 1] = p[2 * xo + 1] + dy * dt;
        }
    }
}

// B6: Streakline::runLK bookkeeping (Streakline.cpp:22-71) with the vertices moved
// through the dense field: next = v + bilinear(flow, v)*dt; vertices the sampler rejects
// keep their position.  Jump rejection Streakline.cpp:35-40 uses the frame size.
void orc_streakline_step(float* verts, int* nverts, float gen_x, float gen_y,
                         const float* flow, size_t flow_step, int w, int h, float dt,
                         int* frame_count) {
    int n = *nverts;
    std::vector<float> next(verts, verts + 2 * n);
    for (int i = 0; i < n; i++) {
        float dx, dy;
        if (sample_flow(flow, flow_step, w, h, verts[2 * i], verts[2 * i + 1], dx, dy)) {
            next[2 * i] = verts[2 * i] + dx * dt;
            next[2 * i + 1] = verts[2 * i + 1] + dy * dt;
        }
        if (std::fabs(verts[2 * i] - next[2 * i]) > w * 0.1 ||
            std::fabs(verts[2 * i + 1] - next[2 * i + 1]) > h * 0.1) {
            next[2 * i] = verts[2 * i];
            next[2 * i + 1] = verts[2 * i + 1];
        }
    }
    // vertices.insert(vertices.begin(), generationPoint)  (frameCount % 1 == 0 always)
    verts[0] = gen_x;
    verts[1] = gen_y;
    std::memcpy(verts + 2, next.data(), sizeof(float) * 2 * n);
    *nverts = n + 1;
    (*frame_count)++;
}

// B7: subtructAverage ripcurrents_module.cpp:810-898 (cv::mean = double sums)
void orc_subtract_average(float* flow, size_t flow_step, int w, int h) {
    double sx = 0, sy = 0;
    for (int y = 0; y < h; y++) {
        const float* f = row_ptr(flow, flow_step, y);
        for (int x = 0; x < w; x++) { sx += f[2 * x]; sy += f[2 * x + 1]; }
    }
    double ax = sx / ((double)w * h), ay = sy / ((double)w * h);
    for (int y = 0; y < h; y++) {
        float* f = row_ptr(flow, flow_step, y);
        for (int x = 0; x < w; x++) {
            f[2 * x] = (float)(f[2 * x] - ax);
            f[2 * x + 1] = (float)(f[2 * x + 1] - ay);
        }
    }
}

// B7: subtructMeanMagnitude ripcurrents_module.cpp:900-1015 (float running sum)
void orc_subtract_mean_magnitude(float* flow, size_t flow_step, int w, int h) {
    float meanval = 0;
    for (int y = 0; y < h; y++) {
        const float* f = row_ptr(flow, flow_step, y);
        for (int x = 0; x < w; x++)
            meanval += std::sqrt(f[2 * x] * f[2 * x] + f[2 * x + 1] * f[2 * x + 1]);
    }
    meanval = meanval / (h * w);
    for (int y = 0; y < h; y++) {
        float* f = row_ptr(flow, flow_step, y);
        for (int x = 0; x < w; x++) {
            float magnitude = std::sqrt(f[2 * x] * f[2 * x] + f[2 * x + 1] * f[2 * x + 1]);
            float ux, uy;
            if (magnitude == 0) { ux = 0.0; uy = 0.0; }
            else { ux = f[2 * x] / magnitude; uy = f[2 * x + 1] / magnitude; }
            f[2 * x] = ux * (magnitude - meanval);
            f[2 * x + 1] = uy * (magnitude - meanval);
        }
    }
}

// B7: stabilizer ripcurrents_module.cpp:279-308 (divides by patch cols / rows, as there)
void orc_stabilizer(float* flow, size_t flow_step, int w, int h) {
    double sum_x = 0, sum_y = 0;
    int cx = 0, cy = 0;
    for (int row = (int)(h * 0.9); row < h; row++) {
        const float* p = row_ptr(flow, flow_step, row) + 2 * (int)(w * 0.9);
        cy++;
        cx = 0;
        for (int col = (int)(w * 0.9); col < w; col++) {
            sum_x += p[0];
            sum_y += p[1];
            cx++;
            p += 2;
        }
    }
    double mean_x = sum_x / cx, mean_y = sum_y / cy;
    for (int row = 0; row < h; row++) {
        float* p = row_ptr(flow, flow_step, row);
        for (int col = 0; col < w; col++) {
            if (p[0] != 0) p[0] = (float)(p[0] - mean_x * 0.2);
            if (p[1] != 0) p[1] = (float)(p[1] - mean_y * 0.2);
            p += 2;
        }
    }
}

// B7: sliding-window mean main.cpp:1142-1153; Mat/float = Mat*(float)(1./float)
void orc_window_mean_update(float* avg, float* slot, const float* cur, int n, int window) {
    float inv = (float)(1. / (float)window);
    for (int i = 0; i < n; i++) {
        float t = slot[i] * inv;
        avg[i] = avg[i] - t;
        slot[i] = cur[i];
        t = slot[i] * inv;
        avg[i] = avg[i] + t;
    }
}

// B8: vectorToColor ripcurrents_module.cpp:1017-1057 (HSV triple before cvtColor).
// *max_displacement is the function-static carried across calls.
void orc_vector_to_color(const float* flow, size_t flow_step, int w, int h, uint8_t* hsv,
                         size_t hsv_step, float* max_displacement) {
    float max_new = 0;
    for (int row = 0; row < h; row++) {
        const float* p = row_ptr(flow, flow_step, row);
        uint8_t* q = row_ptr(hsv, hsv_step, row);
        for (int col = 0; col < w; col++) {
            // ripcurrents_module.cpp:1030: `atan2(ptr->y, ptr->x)` on floats is libm's atan2f, accurate to an ulp but not
            // the same ulp on every platform (glibc 2.35, newer glibc, Apple's libm).  Pinned here as the correctly rounded
            // float: the double atan2 rounded once.  Everything after it is IEEE arithmetic as written.
            float theta = (float)((float)std::atan2((double)p[1], (double)p[0]) * 180 / M_PI);
            theta += theta < 0 ? 360 : 0;
            q[0] = f2u8(theta / 2);
            q[1] = 255;
            float mag = std::sqrt(p[0] * p[0] + p[1] * p[1]);
            q[2] = f2u8(mag * 255 / *max_displacement);
            if (mag > max_new) max_new = mag;
            p += 2;
            q += 3;
        }
    }
    *max_displacement = max_new;
}

// B8: shearRateToColor ripcurrents_module.cpp:1059-1138; interior pixels only.
void orc_shear_rate_to_color(const float* flow, size_t flow_step, int w, int h,
                             uint8_t* hsv, size_t hsv_step, float* max_frobenius) {
    const int offset = 10;
    float max_new = 0.0;
    for (int row = offset; row < h - offset; row++) {
        uint8_t* q = row_ptr(hsv, hsv_step, row) + 3 * offset;
        const float* above = row_ptr(flow, flow_step, row - offset);
        const float* below = row_ptr(flow, flow_step, row + offset);
        const float* mid = row_ptr(flow, flow_step, row);
        for (int col = offset; col < w - offset; col++) {
            float j00 = mid[2 * (col + offset)] - mid[2 * (col - offset)];
            float j01 = above[2 * col] - below[2 * col];
            float j10 = mid[2 * (col + offset) + 1] - mid[2 * (col - offset) + 1];
            float j11 = above[2 * col + 1] - below[2 * col + 1];
            float fro = j00 * j00 + j01 * j01 + j10 * j10 + j11 * j11;
            fro = std::sqrt(fro);
            q[0] = f2u8(128 - fro * 128 / *max_frobenius);
            q[1] = 255;
            q[2] = 255;
            max_new = std::max(fro, max_new);
            q += 3;
        }
    }
    *max_frobenius = max_new;
}

// ---------------------------------------------------------------- section 8(f) next rows
// getStructuringElement(MORPH_ELLIPSE, Size(5,5)) -- imgproc morph.cpp
void orc_ellipse5(uint8_t kernel[25]) {
    const int ksize = 5, r = 2, c = 2;
    double inv_r2 = r ? 1. / ((double)r * r) : 0;
    for (int i = 0; i < ksize; i++) {
        int j1 = 0, j2 = 0;
        int dy = i - r;
        if (std::abs(dy) <= r) {
            int dx = (int)std::nearbyint(c * std::sqrt((r * r - dy * dy) * inv_r2));   // saturate_cast<int> = cvRound
            j1 = std::max(c - dx, 0);
            j2 = std::min(c + dx + 1, ksize);
        }
        for (int j = 0; j < ksize; j++) kernel[i * ksize + j] = (j >= j1 && j < j2) ? 1 : 0;
    }
}

static void morph5(const uint8_t* src, size_t sstep, int w, int h, uint8_t* dst, size_t dstep, const uint8_t* k,
                   bool dilate) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = dilate ? 0 : 255;       // the constant border value that never wins
            for (int i = 0; i < 5; i++)
                for (int j = 0; j < 5; j++) {
                    if (!k[i * 5 + j]) continue;
                    int yy = y + i - 2, xx = x + j - 2;
                    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                    int s = src[yy * sstep + xx];
                    v = dilate ? std::max(v, s) : std::min(v, s);
                }
            dst[y * dstep + x] = (uint8_t)v;
        }
}

// create_edges: dilate(outmask); morphologyEx(outmask, MORPH_GRADIENT) = dilate - erode
void orc_create_edges(const uint8_t* mask, size_t mask_step, int w, int h, uint8_t* out, size_t out_step) {
    uint8_t k[25];
    orc_ellipse5(k);
    std::vector<uint8_t> m1((size_t)w * h), d((size_t)w * h), e((size_t)w * h);
    morph5(mask, mask_step, w, h, m1.data(), w, k, true);
    morph5(m1.data(), w, w, h, d.data(), w, k, true);
    morph5(m1.data(), w, w, h, e.data(), w, k, false);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) out[y * out_step + x] = (uint8_t)(d[(size_t)y * w + x] - e[(size_t)y * w + x]);
}

// resize 8UC3 INTER_LINEAR (resize.cpp: 11-bit fixed-point coefficients, HResizeLinear then the
// uchar VResizeLinear formula) + cvtColor BGR2GRAY (14-bit fixed point)
void orc_resize_bgr_to_gray(const uint8_t* bgr, size_t step, int sw, int sh, uint8_t* gray, size_t gray_step, int dw,
                            int dh) {
    const int cn = 3, COEF = 2048;   // INTER_RESIZE_COEF_SCALE
    double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> alpha(2 * dw), beta(2 * dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = (short)std::nearbyint((1.f - fx) * COEF);      // saturate_cast<short>
        alpha[2 * dx + 1] = (short)std::nearbyint(fx * COEF);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)std::floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[2 * dy] = (short)std::nearbyint((1.f - fy) * COEF);
        beta[2 * dy + 1] = (short)std::nearbyint(fy * COEF);
    }
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = std::min(std::max(yofs[dy], 0), sh - 1), sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
        const uint8_t* S0 = bgr + (size_t)sy0 * step;
        const uint8_t* S1 = bgr + (size_t)sy1 * step;
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
            int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            int px[3];
            for (int c = 0; c < cn; c++) {
                int h0 = S0[sx * cn + c] * a0 + S0[sx1 * cn + c] * a1;
                int h1 = S1[sx * cn + c] * a0 + S1[sx1 * cn + c] * a1;
                px[c] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                px[c] = std::min(std::max(px[c], 0), 255);
            }
            gray[(size_t)dy * gray_step + dx] = (uint8_t)((px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + (1 << 13)) >> 14);
        }
    }
}

// resize 8UC3 INTER_AREA (the first frame: ripcurrents.cpp:186, main.cpp:126 ...) + cvtColor BGR2GRAY.
// imgproc resize.cpp: integer scale factors take resizeAreaFast_ (sum of the area; 2 x 2 through the SIMD
// path (sum + 2) >> 2, otherwise saturate_cast(sum * (1.f / area))); anything else takes resizeArea_ with
// the DecimateAlpha tables of computeResizeAreaTab (float accumulation, row buffer then column sum).
static int area_tab(int ssize, int dsize, double scale, std::vector<int>& si, std::vector<int>& di,
                    std::vector<float>& alpha) {
    si.clear(); di.clear(); alpha.clear();
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cellWidth = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) { di.push_back(dx); si.push_back(sx1 - 1); alpha.push_back((float)((sx1 - fsx1) / cellWidth)); }
        for (int sx = sx1; sx < sx2; sx++) { di.push_back(dx); si.push_back(sx); alpha.push_back((float)(1.0 / cellWidth)); }
        if (fsx2 - sx2 > 1e-3) {
            di.push_back(dx); si.push_back(sx2);
            alpha.push_back((float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth));
        }
    }
    return (int)di.size();
}

void orc_resize_area_bgr_to_gray(const uint8_t* bgr, size_t step, int sw, int sh, uint8_t* gray, size_t gray_step,
                                 int dw, int dh) {
    const int cn = 3;
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    std::vector<uint8_t> res((size_t)dw * dh * cn);
    int iscale_x = (int)std::nearbyint(scale_x), iscale_y = (int)std::nearbyint(scale_y);   // saturate_cast<int>
    bool is_area_fast = std::abs(scale_x - iscale_x) < DBL_EPSILON && std::abs(scale_y - iscale_y) < DBL_EPSILON;
    if (scale_x >= 1 && scale_y >= 1 && is_area_fast) {
        const int area = iscale_x * iscale_y;
        const float scale = 1.f / area;
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    int sum = 0;
                    for (int ky = 0; ky < iscale_y; ky++)
                        for (int kx = 0; kx < iscale_x; kx++)
                            sum += bgr[(size_t)(dy * iscale_y + ky) * step + (size_t)(dx * iscale_x + kx) * cn + c];
                    int v = (iscale_x == 2 && iscale_y == 2) ? (sum + 2) >> 2 : (int)std::nearbyintf(sum * scale);
                    res[((size_t)dy * dw + dx) * cn + c] = (uint8_t)std::min(std::max(v, 0), 255);
                }
    } else {
        // (scale < 1 in either direction is INTER_LINEAR-like upstream; the reference only shrinks)
        std::vector<int> xsi, xdi, ysi, ydi;
        std::vector<float> xa, ya;
        int xn = area_tab(sw, dw, scale_x, xsi, xdi, xa), yn = area_tab(sh, dh, scale_y, ysi, ydi, ya);
        std::vector<float> buf((size_t)dw * cn), sum((size_t)dw * cn, 0.f);
        int prev_dy = ydi[0];
        auto flush = [&](int dy) {
            for (int i = 0; i < dw * cn; i++) {
                float v = sum[i];
                int q = std::fabs(v) < 2147483648.f ? (int)std::nearbyintf(v) : INT32_MIN;
                res[(size_t)dy * dw * cn + i] = (uint8_t)std::min(std::max(q, 0), 255);
            }
        };
        for (int j = 0; j < yn; j++) {
            const float beta = ya[j];
            const int dy = ydi[j];
            const uint8_t* S = bgr + (size_t)ysi[j] * step;
            std::fill(buf.begin(), buf.end(), 0.f);
            for (int k = 0; k < xn; k++)
                for (int c = 0; c < cn; c++) buf[xdi[k] * cn + c] += S[xsi[k] * cn + c] * xa[k];
            if (dy != prev_dy) {
                flush(prev_dy);
                for (int i = 0; i < dw * cn; i++) sum[i] = beta * buf[i];
                prev_dy = dy;
            } else {
                for (int i = 0; i < dw * cn; i++) sum[i] += beta * buf[i];
            }
        }
        flush(prev_dy);
    }
    for (int dy = 0; dy < dh; dy++)
        for (int dx = 0; dx < dw; dx++) {
            const uint8_t* px = res.data() + ((size_t)dy * dw + dx) * cn;
            gray[(size_t)dy * gray_step + dx] = (uint8_t)((px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + (1 << 13)) >> 14);
        }
}

}  // extern "C"
