// oracle_kat.cpp -- drives every entry point of the CPU oracle on small deterministic inputs (ragged sizes, all five
// reference parameter sets, threads, diagnostics, the analysis rows, LK, display) and prints one FNV-1a hash per
// result.  TEST INFRASTRUCTURE: built twice by oracle/Makefile -- with the oracle's normal flags and with
// -fsanitize=address,undefined -- and run by tests/test_oracle_sanitizers.py, which requires a clean sanitizer run
// and identical hashes from both builds (no result depends on undefined behaviour or on the optimisation level).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "rc_oracle.h"

static uint64_t fnv(const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
template <class T> static void show(const char* name, const std::vector<T>& v) {
    printf("%-34s %016llx\n", name, (unsigned long long)fnv(v.data(), v.size() * sizeof(T)));
}

static void frame(std::vector<uint8_t>& f, int w, int h, int t) {
    f.resize((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double u = x - 1.25 * t, v = y + 0.75 * t;
            double s = 128 + 40 * std::sin(u / 7.0) * std::cos(v / 9.0) + 30 * std::sin((u + v) / 13.0) + 20 * std::cos(u / 3.1 - v / 4.3);
            f[(size_t)y * w + x] = (uint8_t)std::lrint(std::fmin(255.0, std::fmax(0.0, s)));
        }
}

int main() {
    struct P { double ps; int lv, win, it, n; double sg; int fl; } sets[] = {
        {0.5, 2, 3, 2, 15, 1.2, 0}, {0.5, 2, 3, 2, 15, 1.2, 256}, {0.5, 2, 20, 3, 15, 1.2, 256}, {0.5, 2, 10, 3, 15, 1.2, 256},
        {0.5, 3, 5, 3, 15, 1.2, 0}, {0.8, 3, 15, 3, 5, 1.1, 0}, {0.5, 2, 1, 1, 5, 1.1, 0}, {0.5, 0, 4, 0, 7, 1.5, 256}};
    const int sizes[][2] = {{97, 70}, {64, 33}, {40, 36}, {2, 200}, {1, 1}, {131, 66}};
    char name[96];
    for (auto& sz : sizes) {
        const int w = sz[0], h = sz[1];
        std::vector<uint8_t> a, b;
        frame(a, w, h, 0); frame(b, w, h, 1);
        int si = 0;
        for (auto& p : sets) {
            std::vector<float> flow((size_t)w * h * 2), flow2(flow.size()), g((size_t)w * h * 3), dm((size_t)w * h);
            if (orc_farneback_u8(a.data(), w, b.data(), w, w, h, flow.data(), (size_t)w * 8, p.ps, p.lv, p.win, p.it, p.n, p.sg, p.fl, 1)) return 2;
            orc_farneback_diag d;
            memset(&d, 0, sizeof(d));
            d.g_last = g.data(); d.det_min = dm.data();
            std::vector<std::vector<float>> lf(ORC_MAX_DIAG_LEVELS);
            int L = orc_level_geometry(w, h, p.ps, p.lv, 0, nullptr, nullptr, nullptr, nullptr);
            for (int k = 0; k <= L; k++) {
                int wk, hk;
                orc_level_geometry(w, h, p.ps, p.lv, k, &wk, &hk, nullptr, nullptr);
                lf[k].assign((size_t)wk * hk * 2, 0.f);
                d.level_flow[k] = lf[k].data();
            }
            if (orc_farneback_u8_ex(a.data(), w, b.data(), w, w, h, flow2.data(), (size_t)w * 8, p.ps, p.lv, p.win, p.it, p.n, p.sg, p.fl, 3, &d)) return 3;
            if (memcmp(flow.data(), flow2.data(), flow.size() * 4)) { printf("threads / diagnostics change the flow\n"); return 4; }
            snprintf(name, sizeof(name), "farneback %dx%d set %d", w, h, si++);
            show(name, flow);
        }
    }
    {   // analysis rows on a flow field with edge values
        const int w = 96, h = 80;
        std::vector<uint8_t> a, b;
        frame(a, w, h, 3); frame(b, w, h, 4);
        std::vector<float> flow((size_t)w * h * 2);
        orc_farneback_u8(a.data(), w, b.data(), w, w, h, flow.data(), (size_t)w * 8, 0.5, 2, 3, 2, 15, 1.2, 0, 1);
        flow[0] = NAN; flow[5] = INFINITY; flow[8] = 3e38f; flow[11] = -3e38f; flow[14] = 2.5f; flow[15] = 0.f;   // hazards
        std::vector<float> polar((size_t)w * h * 3), wc(polar.size(), 0.f), acc2(polar.size(), 0.f), acc(polar.size(), 0.f), out(polar.size(), 0.f);
        std::vector<uint8_t> mask((size_t)w * h), edges(mask.size());
        std::vector<int32_t> hist(ORC_HIST_BINS, 0), hist2d(ORC_HIST_BINS * ORC_HIST_DIRECTIONS, 0), hs2d(ORC_HIST_DIRECTIONS, 0);
        int32_t hs = 0;
        float U = 100.f;
        std::vector<float> U2(ORC_HIST_DIRECTIONS), prop(ORC_HIST_DIRECTIONS);
        orc_flow_to_polar(flow.data(), (size_t)w * 8, w, h, polar.data(), (size_t)w * 12);
        orc_histogram_accumulate(polar.data(), (size_t)w * 12, w, h, hist.data(), &hs, hist2d.data(), hs2d.data());
        orc_histogram_thresholds(hist.data(), hs, hist2d.data(), hs2d.data(), &U, U2.data(), prop.data());
        show("hist2d", hist2d); show("UPPER2d", U2);
        orc_create_flow(polar.data(), (size_t)w * 12, wc.data(), (size_t)w * 12, acc2.data(), (size_t)w * 12, w, h, U, 0.5f, 0.2f, U2.data());
        orc_create_accumulationbuffer(acc.data(), (size_t)w * 12, acc2.data(), (size_t)w * 12, out.data(), (size_t)w * 12, mask.data(), w, w, h, 31);
        show("outmask", mask);
        orc_create_edges(mask.data(), w, w, h, edges.data(), w);
        show("edges", edges);
        std::vector<float> pt((size_t)w * h * 2, 0.f), dist((size_t)w * h, 0.f);
        for (int it = 0; it < 3; it++) orc_streamline_field(pt.data(), (size_t)w * 8, dist.data(), (size_t)w * 4, flow.data(), (size_t)w * 8, w, h, 2.f, 1, U);
        show("streamline_field pt", pt);
        std::vector<float> seeds = {10.f, 10.f, 0.2f, 0.3f, 95.f, 79.f, 50.5f, 40.25f, 3e38f, 1.f, NAN, 5.f};
        for (int v = 0; v < 5; v++) {
            std::vector<float> s = seeds, tr((size_t)6 * 100 * 2, 0.f);
            orc_streamline_points(s.data(), 6, flow.data(), (size_t)w * 8, w, h, 0.5f, 7, 100.f, v, tr.data());
            snprintf(name, sizeof(name), "streamline variant %d", v);
            show(name, tr);
        }
        std::vector<float> verts(2 * 16, 0.f);
        verts[0] = 48.f; verts[1] = 40.f;
        int nv = 1, fc = 1;
        for (int it = 0; it < 6; it++) orc_streakline_step(verts.data(), &nv, 48.f, 40.f, flow.data(), (size_t)w * 8, w, h, 1.f, &fc);
        show("streakline", verts);
        std::vector<uint8_t> img((size_t)w * h * 3);
        double mx;
        for (int which = 0; which < 3; which++) {
            orc_streamline_display(pt.data(), (size_t)w * 8, dist.data(), (size_t)w * 4, w, h, which, img.data(), (size_t)w * 3, &mx);
            snprintf(name, sizeof(name), "display %d", which);
            show(name, img);
        }
        std::vector<float> bgr(polar.size());
        orc_hsv_to_bgr_f32(polar.data(), (size_t)w * 12, w, h, bgr.data(), (size_t)w * 12);
        std::vector<float> f2 = flow;
        f2[0] = 0.f; f2[5] = 0.f; f2[8] = 1.f; f2[11] = -1.f;
        std::vector<uint8_t> hsv((size_t)w * h * 3);
        float md = 1.f, mf = 1.f;
        orc_vector_to_color(f2.data(), (size_t)w * 8, w, h, hsv.data(), (size_t)w * 3, &md);
        show("vector_to_color", hsv);
        orc_shear_rate_to_color(f2.data(), (size_t)w * 8, w, h, hsv.data(), (size_t)w * 3, &mf);
        show("shear_rate_to_color", hsv);
        orc_subtract_average(f2.data(), (size_t)w * 8, w, h);
        orc_subtract_mean_magnitude(f2.data(), (size_t)w * 8, w, h);
        orc_stabilizer(f2.data(), (size_t)w * 8, w, h);
        show("post-ops", f2);
        std::vector<uint8_t> bgr8((size_t)211 * 133 * 3), gray((size_t)w * h);
        for (size_t i = 0; i < bgr8.size(); i++) bgr8[i] = (uint8_t)((i * 2654435761u) >> 24);
        orc_resize_bgr_to_gray(bgr8.data(), 211 * 3, 211, 133, gray.data(), w, w, h);
        show("resize linear", gray);
        orc_resize_area_bgr_to_gray(bgr8.data(), 211 * 3, 211, 133, gray.data(), w, w, h);
        show("resize area", gray);
    }
    printf("oracle_kat: ok\n");
    return 0;
}
