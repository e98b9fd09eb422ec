/*
 * display_oracle.cpp -- CPU restatement of the flow display path (TEST INFRASTRUCTURE, NOT PRODUCT).
 * ripcurrents.cpp:233-273 (= ripcurrents_module.cpp:13-60:
 * streamline_displacement / _total_motion / _ratio / _positions) and ripcurrents.cpp:405
 * (cvtColor(current, CV_HSV2BGR) on the 32FC3 display image).
 *
 * PARITY UNPINNED: minMaxLoc, convertTo, divide, magnitude, applyColorMap(COLORMAP_JET) and the float
 * HSV->BGR conversion live in un-vendored OpenCV 4.1.0 (core, imgproc colormap.cpp / color_hsv.cpp).
 * Restated here: JET = the 256 tabulated control points of colormap.cpp (clip(1.5 - |4x - c|)), passed
 * through its float linear_colormap / interp1 and scaled by 255 with round-half-even; convertTo(8U)
 * = saturate(cvRound(v * (float)alpha)); HSV2RGB_f's sector table.
 */
#include "rc_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

inline int cv_round_f(float v) {
    if (!(std::fabs(v) < 2147483648.f)) return INT32_MIN;      // cvtss2si on NaN / overflow
    return (int)std::nearbyintf(v);
}
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// colormap.cpp Jet: 256 tabulated control points per channel, the classic
// clip(1.5 - |4x - c|, 0, 1) with x = i/255 and c = 3 (r), 2 (g), 1 (b), written there as decimal
// literals (0.5, 0.5156862745098039, ... / 0.00588235294117645, 0.02156862745098032, ...):
// the float nearest to the double value.
void jet256(float r[256], float g[256], float b[256]) {
    for (int i = 0; i < 256; i++) {
        double x = i / 255.0;
        auto f = [](double v) { return (float)(v < 0 ? 0 : (v > 1 ? 1 : v)); };
        r[i] = f(1.5 - std::fabs(4 * x - 3));
        g[i] = f(1.5 - std::fabs(4 * x - 2));
        b[i] = f(1.5 - std::fabs(4 * x - 1));
    }
}

// colormap.cpp interp1 (float)
float interp(const float* X, const float* Y, int nx, float xi) {
    int low = 0, high = nx - 1;
    if (xi < X[low]) high = 1;
    if (xi > X[high]) low = high - 1;
    while (high - low > 1) {
        int c = low + ((high - low) >> 1);
        if (xi > X[c]) low = c; else high = c;
    }
    return Y[low] + (xi - X[low]) * (Y[high] - Y[low]) / (X[high] - X[low]);
}

}  // namespace

extern "C" void orc_jet_lut(uint8_t* lut_bgr /* 256*3 */) {
    float r[256], g[256], b[256], X[256];
    jet256(r, g, b);
    const float step = (1.f - 0.f) / (256 - 1);          // linspace(0, 1, 256), both X and XI
    for (int i = 0; i < 256; i++) X[i] = 0.f + i * step;
    for (int i = 0; i < 256; i++) {
        float xi = X[i];
        lut_bgr[3 * i + 0] = sat_u8(cv_round_f(interp(X, b, 256, xi) * 255.f));
        lut_bgr[3 * i + 1] = sat_u8(cv_round_f(interp(X, g, 256, xi) * 255.f));
        lut_bgr[3 * i + 2] = sat_u8(cv_round_f(interp(X, r, 256, xi) * 255.f));
    }
}

/* which: 0 = streamline_displacement (|pt|), 1 = streamline_total_motion (dist), 2 = streamline_ratio
 * (|pt| / dist).  bgr is h x w x 3 8-bit; *max_out receives the minMaxLoc maximum. */
extern "C" void orc_streamline_display(const float* pt, size_t pt_step, const float* dist, size_t dist_step, int w,
                                       int h, int which, uint8_t* bgr, size_t bgr_step, double* max_out) {
    std::vector<float> v((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const float* p = (const float*)((const char*)pt + (size_t)y * pt_step);
        const float* d = (const float*)((const char*)dist + (size_t)y * dist_step);
        for (int x = 0; x < w; x++) {
            float mag = std::sqrt(p[2 * x] * p[2 * x] + p[2 * x + 1] * p[2 * x + 1]);     // cv::magnitude
            v[(size_t)y * w + x] = which == 0 ? mag : which == 1 ? d[x] : mag / d[x];   // cv::divide: IEEE
        }
    }
    double mx = -INFINITY;      // minMaxLoc: NaNs never compare greater
    for (size_t i = 0; i < v.size(); i++) if (v[i] > mx) mx = v[i];
    if (max_out) *max_out = mx;
    const float alpha = (float)(255 / mx);
    uint8_t lut[768];
    orc_jet_lut(lut);
    for (int y = 0; y < h; y++) {
        uint8_t* o = bgr + (size_t)y * bgr_step;
        for (int x = 0; x < w; x++) {
            int idx = sat_u8(cv_round_f(v[(size_t)y * w + x] * alpha));
            o[3 * x] = lut[3 * idx]; o[3 * x + 1] = lut[3 * idx + 1]; o[3 * x + 2] = lut[3 * idx + 2];
        }
    }
}

/* streamline_positions ripcurrents_module.cpp:44-60: marks where each pixel's particle sits */
extern "C" void orc_streamline_positions(const float* pt, size_t pt_step, int w, int h, float* density,
                                         size_t density_step) {
    for (int y = 0; y < h; y++) {
        const float* p = (const float*)((const char*)pt + (size_t)y * pt_step);
        for (int x = 0; x < w; x++) {
            int xind = (int)roundf(std::floor(p[2 * x] + x));
            int yind = (int)roundf(std::floor(p[2 * x + 1] + y));
            if (xind < 1 || yind < 1 || xind + 2 > w || yind + 2 > h) continue;
            float* d = (float*)((char*)density + (size_t)yind * density_step) + 3 * xind;
            d[0] = d[1] = d[2] = 1.f;
        }
    }
}

/* cvtColor(32FC3, CV_HSV2BGR) imgproc color_hsv.cpp HSV2RGB_f (H in degrees), ripcurrents.cpp:405 */
extern "C" void orc_hsv_to_bgr_f32(const float* hsv, size_t hsv_step, int w, int h, float* bgr, size_t bgr_step) {
    static const int sector_data[][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    const float hscale = 6.f / 360.f;
    for (int y = 0; y < h; y++) {
        const float* s = (const float*)((const char*)hsv + (size_t)y * hsv_step);
        float* d = (float*)((char*)bgr + (size_t)y * bgr_step);
        for (int x = 0; x < w; x++) {
            float hh = s[3 * x], ss = s[3 * x + 1], vv = s[3 * x + 2], b, g, r;
            if (ss == 0) b = g = r = vv;
            else {
                float tab[4];
                hh *= hscale;
                // color_hsv.cpp wraps the hue with `do h -= 6; while (h >= 6)`, which never ends for an infinite or
                // huge hue (x - 6 == x).  Same steps here, at most 64 of them (|hue| < 3840 degrees behaves like
                // upstream); a hue still out of range after that is taken as 0 instead of hanging.
                for (int it = 0; it < 64 && hh < 0; it++) hh += 6;
                for (int it = 0; it < 64 && hh >= 6; it++) hh -= 6;
                if (hh < 0 || hh >= 6) hh = 0.f;
                int sector = (int)std::floor(hh);
                hh -= sector;
                if ((unsigned)sector >= 6u) { sector = 0; hh = 0.f; }
                tab[0] = vv;
                tab[1] = vv * (1.f - ss);
                tab[2] = vv * (1.f - ss * hh);
                tab[3] = vv * (1.f - ss * (1.f - hh));
                b = tab[sector_data[sector][0]];
                g = tab[sector_data[sector][1]];
                r = tab[sector_data[sector][2]];
            }
            d[3 * x] = b; d[3 * x + 1] = g; d[3 * x + 2] = r;
        }
    }
}
