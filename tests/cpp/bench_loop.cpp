// The reference's own loop shape, measured from a C++ host with no Python anywhere: one frame per call
// (ripcurrents.cpp:194-221: video.read -> resize -> cvtColor -> copyTo(UMat) -> calcOpticalFlowFarneback -> copyTo(u_f2))
// followed by the per-frame analysis chain of ripcurrents.cpp:229-479 on the resident flow field.
//
//   bench_loop [W H [frames]]          default 640 480 600 (ripcurrents.hpp:4-5), then 1920 1080
//
// Modes, one JSON line each:
//   flow        frame produced into the slot's page-locked staging buffer (rcflow_frame_buffer_acquire: the cvtColor
//               destination), rcflow_push_frame_acquired; the flow stays on the device
//   loop        flow + streamline_field (:229-231) + 250 seed streamlines (:283-285) + histogram and thresholds
//               (:319-366) + classify / accumulate (:376-439) + mask edges (:477-479), every call through the C ABI
//   loop-graph  the same chain as ONE hipGraph launch per frame (rcflow_frame_loop_step): same bits as `loop`
//   loop-step   rcflow_frame_loop_step without the graph: the same launches as `loop` behind one call per frame
// Frames come from a synthetic clip generated in host memory before the clock starts (the decode is the host's
// business); everything else -- the copy into the staging buffer, the PCIe upload, every kernel -- is inside the
// timed region, which ends with one synchronisation after the last frame.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rcflow.h"

#define CK(x) do { int rc_ = (x); if (rc_ < 0) { printf("FAILED %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, rcflow_last_error()); exit(1); } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP FAILED %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static void make_clip(std::vector<uint8_t>& clip, int w, int h, int T) {
    clip.resize((size_t)w * h * T);
    for (int t = 0; t < T; t++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                double u = x - 1.25 * t, v = y + 0.75 * t;
                double s = 128 + 40 * std::sin(u / 7.0) * std::cos(v / 9.0) + 30 * std::sin((u + v) / 13.0) + 20 * std::cos(u / 3.1 - v / 4.3);
                clip[((size_t)t * h + y) * w + x] = (uint8_t)std::lrint(std::fmin(255.0, std::fmax(0.0, s)));
            }
}

struct Loop {
    rc_ctx* ctx = nullptr;
    int w, h;
    float* d_seeds = nullptr;
    uint8_t *d_mask = nullptr, *d_edges = nullptr;
    rc_farneback_params prm = {0.5, 2, 3, 2, 15, 1.2, 0};      // ripcurrents.cpp:215
    std::vector<float> seeds0;
    Loop(int w_, int h_) : w(w_), h(h_) {
        CK(rcflow_create(&ctx, 0, w, h, 1));
        CK(rcflow_analysis_reset(ctx, 0, w, h));
        seeds0.resize(500);
        unsigned s = 12345;
        for (int i = 0; i < 250; i++) {                       // ripcurrents.cpp:170-172: 250 random seeds
            s = s * 1664525u + 1013904223u; seeds0[2 * i] = (float)(s >> 8) / 16777216.f * w;
            s = s * 1664525u + 1013904223u; seeds0[2 * i + 1] = (float)(s >> 8) / 16777216.f * h;
        }
        HK(hipMalloc((void**)&d_seeds, 2000));
        HK(hipMalloc((void**)&d_mask, (size_t)w * h));
        HK(hipMalloc((void**)&d_edges, (size_t)w * h));
    }
    ~Loop() { (void)hipFree(d_seeds); (void)hipFree(d_mask); (void)hipFree(d_edges); rcflow_destroy(ctx); }
    void restart() {
        CK(rcflow_sync(ctx, 0));
        CK(rcflow_stream_reset(ctx, 0));
        CK(rcflow_analysis_reset(ctx, 0, w, h));
        HK(hipMemcpy(d_seeds, seeds0.data(), 2000, hipMemcpyHostToDevice));
        HK(hipMemset(d_mask, 0, (size_t)w * h));
        HK(hipMemset(d_edges, 0, (size_t)w * h));
    }
    // returns true when a flow field exists (every frame but the first)
    bool produce_and_push(const uint8_t* frame) {
        uint8_t* buf = nullptr;
        size_t step = 0;
        CK(rcflow_frame_buffer_acquire(ctx, 0, w, h, &buf, &step));
        for (int y = 0; y < h; y++) memcpy(buf + (size_t)y * step, frame + (size_t)y * w, w);   // stands for cvtColor writing its destination
        int rc = rcflow_push_frame_acquired(ctx, 0, &prm);
        CK(rc);
        return rc == 0;
    }
    void analysis(int framecount) {
        float* d_flow = nullptr;
        CK(rcflow_stream_flow_ptr(ctx, 0, &d_flow, nullptr, nullptr));
        const size_t fs = (size_t)w * 8;
        CK(rcflow_advect_field_dev(ctx, 0, d_flow, fs, w, h, 2.f, 1, -1.f));                       // :229-231 (the previous frame's UPPER)
        CK(rcflow_advect_points_dev(ctx, 0, d_seeds, 250, d_flow, fs, w, h, 2.f, 1, 100.f, 3, nullptr));   // :283-285
        CK(rcflow_histogram_dev(ctx, 0, d_flow, fs, w, h));                                      // :319-330
        CK(rcflow_thresholds_dev(ctx, 0));                                                       // :333-366
        CK(rcflow_classify_accumulate_dev(ctx, 0, d_flow, fs, w, h, framecount, 0.5f, 0.2f, nullptr, 0, nullptr, 0, nullptr, 0, d_mask, w));   // :376-439
        CK(rcflow_create_edges_dev(ctx, 0, d_mask, w, w, h, d_edges, w));                          // :477-479
    }
};

static unsigned long long checksum(const void* d, size_t n) {
    std::vector<uint8_t> hbuf(n);
    HK(hipMemcpy(hbuf.data(), d, n, hipMemcpyDeviceToHost));
    unsigned long long s = 1469598103934665603ull;
    for (uint8_t b : hbuf) { s ^= b; s *= 1099511628211ull; }
    return s;
}

static void run(int w, int h, int frames) {
    const int T = 16;
    std::vector<uint8_t> clip;
    make_clip(clip, w, h, T);
    auto frame_at = [&](int i) { int k = i % (2 * T - 2); if (k >= T) k = 2 * T - 2 - k; return clip.data() + (size_t)k * w * h; };   // forwards, then backwards
    Loop L(w, h);
    unsigned long long sums[4][3] = {{0}};
    for (int mode = 0; mode < 4; mode++) {
        const char* name = mode == 0 ? "flow" : (mode == 1 ? "loop" : (mode == 2 ? "loop-graph" : "loop-step"));
        rc_frame_loop cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.dt = 2.f; cfg.iterations = 1; cfg.d_seeds = L.d_seeds; cfg.nseeds = 250; cfg.seed_variant = 3; cfg.seed_upper = 100.f; cfg.seed_dt = 2.f; cfg.seed_iterations = 1;
        cfg.MID = 0.5f; cfg.LOWER = 0.2f; cfg.d_outmask = L.d_mask; cfg.mask_step = w; cfg.d_edges = L.d_edges; cfg.edges_step = w; cfg.use_graph = mode == 2;
        double best = 0;
        for (int rep = 0; rep < 3; rep++) {
            L.restart();
            // untimed: the priming frame and a few frames to allocate, capture and warm up
            int fc = 0;
            for (int i = 0; i < 8; i++) {
                if (mode >= 2) { uint8_t* b; size_t st; CK(rcflow_frame_buffer_acquire(L.ctx, 0, w, h, &b, &st));
                    for (int y = 0; y < h; y++) memcpy(b + (size_t)y * st, frame_at(i) + (size_t)y * w, w);
                    int r2 = rcflow_frame_loop_step(L.ctx, 0, &L.prm, &cfg); CK(r2); if (r2 == 0) fc++; continue; }
                if (L.produce_and_push(frame_at(i))) { fc++; if (mode >= 1) L.analysis(fc); }
            }
            CK(rcflow_sync(L.ctx, 0));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 8; i < 8 + frames; i++) {
                if (mode >= 2) {
                    uint8_t* b; size_t st;
                    CK(rcflow_frame_buffer_acquire(L.ctx, 0, w, h, &b, &st));
                    for (int y = 0; y < h; y++) memcpy(b + (size_t)y * st, frame_at(i) + (size_t)y * w, w);
                    CK(rcflow_frame_loop_step(L.ctx, 0, &L.prm, &cfg));
                    fc++;
                } else if (L.produce_and_push(frame_at(i))) {
                    fc++;
                    if (mode >= 1) L.analysis(fc);
                }
            }
            CK(rcflow_sync(L.ctx, 0));
            double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            best = std::fmax(best, frames / s);
            if (rep == 0) {
                float* d_flow = nullptr;
                CK(rcflow_stream_flow_ptr(L.ctx, 0, &d_flow, nullptr, nullptr));
                sums[mode][0] = checksum(d_flow, (size_t)w * h * 8);
                sums[mode][1] = checksum(L.d_edges, (size_t)w * h);
                sums[mode][2] = checksum(L.d_seeds, 2000);
            }
        }
        printf("{\"metric\": \"frames/sec, one frame per call from a C++ host (reference-shaped loop)\", \"mode\": \"%s\", \"w\": %d, \"h\": %d, "
               "\"frames\": %d, \"value\": %.1f, \"unit\": \"frames/s\", \"us_per_frame\": %.2f, \"flow_fnv\": \"%016llx\", \"edges_fnv\": \"%016llx\", "
               "\"seeds_fnv\": \"%016llx\"}\n", name, w, h, frames, best, 1e6 / best, sums[mode][0], sums[mode][1], sums[mode][2]);
        fflush(stdout);
    }
    // the graph replay must leave the same bits as the eager loop; all three modes the same flow field
    if (sums[1][0] != sums[2][0] || sums[1][1] != sums[2][1] || sums[1][2] != sums[2][2] || sums[0][0] != sums[1][0] ||
        sums[3][0] != sums[1][0] || sums[3][1] != sums[1][1] || sums[3][2] != sums[1][2]) {
        printf("FAILED: the modes disagree (flow / edges / seeds checksums)\n");
        exit(1);
    }
}

int main(int argc, char** argv) {
    if (argc >= 3) { run(atoi(argv[1]), atoi(argv[2]), argc >= 4 ? atoi(argv[3]) : 600); }
    else { run(640, 480, 600); run(1920, 1080, 300); }
    printf("bench_loop ok\n");
    return 0;
}
