// C++ test of the host mirror (include/rcflow_module.hpp) against the CPU oracle, reading like
// the reference's frame loop (ripcurrents.cpp:194-440).  Built by __graft_entry__.build()
// (hipcc), run on the GPU box by tests/test_gpu_cpp_module.py.  Links the oracle as checker.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rcflow_module.hpp"
#include "../../oracle/rc_oracle.h"

#define XDIM 320
#define YDIM 240
#define REQUIRE(c)                                                        \
    do {                                                                  \
        if (!(c)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } \
    } while (0)

static void make_frame(std::vector<uint8_t>& f, int t) {   // moving smooth texture
    f.resize((size_t)XDIM * YDIM);
    for (int y = 0; y < YDIM; y++)
        for (int x = 0; x < XDIM; x++) {
            double u = x - 1.25 * t, v = y + 0.75 * t;
            double s = 128 + 40 * std::sin(u / 7.0) * std::cos(v / 9.0) + 30 * std::sin((u + v) / 13.0) +
                       20 * std::cos(u / 3.1 - v / 4.3);
            f[(size_t)y * XDIM + x] = (uint8_t)std::lrint(std::fmin(255.0, std::fmax(0.0, s)));
        }
}

int main() {
    rc::Pipeline pipe(XDIM, YDIM);
    std::vector<uint8_t> f1, f2;
    make_frame(f2, 0);
    // state of ripcurrents.cpp:133-176
    int hist[RC_HIST_BINS] = {0}, histsum = 0, hist2d[RC_HIST_DIRECTIONS][RC_HIST_BINS] = {{0}}, histsum2d[RC_HIST_DIRECTIONS] = {0};
    float UPPER = 100.0f, UPPER2d[RC_HIST_DIRECTIONS] = {0}, prop[RC_HIST_DIRECTIONS] = {0};
    int ohist[RC_HIST_BINS] = {0}, ohistsum = 0, ohist2d[RC_HIST_DIRECTIONS * RC_HIST_BINS] = {0}, ohistsum2d[RC_HIST_DIRECTIONS] = {0};
    float oUPPER = 100.0f, oUPPER2d[RC_HIST_DIRECTIONS] = {0}, oprop[RC_HIST_DIRECTIONS] = {0};
    std::vector<float> flow((size_t)XDIM * YDIM * 2), oflow(flow.size()), polar((size_t)XDIM * YDIM * 3);
    std::vector<float> opt((size_t)XDIM * YDIM * 2, 0.f), odist((size_t)XDIM * YDIM, 0.f);
    std::vector<float> acc((size_t)XDIM * YDIM * 3, 0.f);
    std::vector<uint8_t> mask((size_t)XDIM * YDIM), omask(mask.size());
    rc::Streakline streak(rc::Pixel2{160.f, 120.f});
    std::vector<float> overts(2 * 64, 0.f);
    overts[0] = 160.f; overts[1] = 120.f;
    int on = 1, ofc = 1;
    rc::Streakline lkstreak(rc::Pixel2{200.f, 90.f});       // the reference's own mover: sparse PyrLK
    std::vector<float> lkverts(2 * 64, 0.f);
    lkverts[0] = 200.f; lkverts[1] = 90.f;
    int lkn = 1, lkfc = 1;
    rc::Timeline timeline(rc::Pixel2{60.f, 60.f}, rc::Pixel2{260.f, 180.f}, 10);
    REQUIRE(timeline.vertices.size() == 11);

    for (int framecount = 1; framecount <= 4; framecount++) {
        make_frame(f1, framecount);
        rc::Mat prev(YDIM, XDIM, 1, 1, f2.data()), next(YDIM, XDIM, 1, 1, f1.data()), mflow(YDIM, XDIM, 2, 4, flow.data());
        pipe.calcOpticalFlowFarneback(prev, next, mflow, 0.5, 2, 3, 2, 15, 1.2, 0);       // ripcurrents.cpp:215
        REQUIRE(orc_farneback_u8(f2.data(), XDIM, f1.data(), XDIM, XDIM, YDIM, oflow.data(), XDIM * 8, 0.5, 2, 3, 2, 15, 1.2, 0, 1) == 0);
        size_t good = 0;
        for (size_t i = 0; i < flow.size(); i += 2)
            good += std::fabs(flow[i] - oflow[i]) <= 1e-3f && std::fabs(flow[i + 1] - oflow[i + 1]) <= 1e-3f;
        REQUIRE(good >= (size_t)(0.99 * XDIM * YDIM));
        timeline.runLK(pipe, prev, next);                                                    // ripcurrents_module.cpp:765-807
        lkstreak.runLK(pipe, prev, next);                                                    // Streakline.cpp:22-71
        REQUIRE(orc_streakline_step_lk(lkverts.data(), &lkn, 200.f, 90.f, f2.data(), XDIM, f1.data(), XDIM, XDIM, YDIM, &lkfc) == 0);
        f2 = f1;                                                                             // u_f1.copyTo(u_f2)

        // from here both sides consume the GPU flow field, so everything must agree exactly
        pipe.streamline_field(2.f, 1);                                                       // ripcurrents.cpp:229-231
        orc_streamline_field(opt.data(), XDIM * 8, odist.data(), XDIM * 4, flow.data(), XDIM * 8, XDIM, YDIM, 2.f, 1, oUPPER);
        if (framecount == 3) {                                                               // ripcurrents.cpp:233-257
            std::vector<uint8_t> img((size_t)XDIM * YDIM * 3), oimg(img.size());
            std::vector<float> spt((size_t)XDIM * YDIM * 2), sdist((size_t)XDIM * YDIM);
            pipe.streamline_field_state((rc::Pixel2*)spt.data(), sdist.data());
            rc::Mat mimg(YDIM, XDIM, 3, 1, img.data());
            for (int which = 0; which < 3; which++) {
                if (which == 0) pipe.streamline_displacement(mimg);
                else if (which == 1) pipe.streamline_total_motion(mimg);
                else pipe.streamline_ratio(mimg);
                double mx;
                orc_streamline_display(spt.data(), XDIM * 8, sdist.data(), XDIM * 4, XDIM, YDIM, which, oimg.data(), XDIM * 3, &mx);
                REQUIRE(img == oimg);
            }
        }
        streak.run(pipe);
        orc_streakline_step(overts.data(), &on, 160.f, 120.f, flow.data(), XDIM * 8, XDIM, YDIM, 1.f, &ofc);

        pipe.create_histogram(hist, histsum, hist2d, histsum2d, UPPER, UPPER2d, prop);       // ripcurrents.cpp:319-366
        orc_flow_to_polar(flow.data(), XDIM * 8, XDIM, YDIM, polar.data(), XDIM * 12);
        orc_histogram_accumulate(polar.data(), XDIM * 12, XDIM, YDIM, ohist, &ohistsum, ohist2d, ohistsum2d);
        orc_histogram_thresholds(ohist, ohistsum, ohist2d, ohistsum2d, &oUPPER, oUPPER2d, oprop);
        REQUIRE(histsum == ohistsum && UPPER == oUPPER);
        for (int d = 0; d < RC_HIST_DIRECTIONS; d++) {
            REQUIRE(histsum2d[d] == ohistsum2d[d] && UPPER2d[d] == oUPPER2d[d]);
            for (int b = 0; b < RC_HIST_BINS; b++) REQUIRE(hist2d[d][b] == ohist2d[d * RC_HIST_BINS + b]);
        }

        rc::Mat mmask(YDIM, XDIM, 1, 1, mask.data());
        pipe.create_flow_and_accumulationbuffer(mmask, framecount + 30);                     // ripcurrents.cpp:376-439
        std::vector<float> wc((size_t)XDIM * YDIM * 3, 0.f), acc2(wc.size(), 0.f), out(wc.size(), 0.f);
        std::fill(omask.begin(), omask.end(), 0);
        orc_create_flow(polar.data(), XDIM * 12, wc.data(), XDIM * 12, acc2.data(), XDIM * 12, XDIM, YDIM, oUPPER, 0.5f, 0.2f, oUPPER2d);
        orc_create_accumulationbuffer(acc.data(), XDIM * 12, acc2.data(), XDIM * 12, out.data(), XDIM * 12, omask.data(), XDIM, XDIM, YDIM, framecount + 30);
        REQUIRE(mask == omask);
    }
    std::vector<rc::Pixel2> pt((size_t)XDIM * YDIM);
    std::vector<float> dist((size_t)XDIM * YDIM);
    pipe.streamline_field_state(pt.data(), dist.data());
    for (size_t i = 0; i < dist.size(); i++) REQUIRE(pt[i].x == opt[2 * i] && pt[i].y == opt[2 * i + 1] && dist[i] == odist[i]);
    REQUIRE(streak.numberOfVertices == on && streak.frameCount == ofc);
    REQUIRE(timeline.vertices.size() == 11 && timeline.vertices[0].x != 60.f);     // every vertex was tracked
    REQUIRE(lkstreak.numberOfVertices == lkn && lkstreak.frameCount == lkfc);
    for (int i = 0; i < lkn; i++)      // exact window sums (GPU) vs raster-order float sums (oracle): ~1e-4 px
        REQUIRE(std::fabs(lkstreak.vertices[i].x - lkverts[2 * i]) < 5e-3f && std::fabs(lkstreak.vertices[i].y - lkverts[2 * i + 1]) < 5e-3f);
    for (int i = 0; i < on; i++) REQUIRE(streak.vertices[i].x == overts[2 * i] && streak.vertices[i].y == overts[2 * i + 1]);

    // error behaviour: bad arguments throw (the reference's cv:: calls throw cv::Exception)
    // the frame loop with the previous frame kept on the device: same flow as the two-image call
    {
        std::vector<uint8_t> a((size_t)XDIM * YDIM), b((size_t)XDIM * YDIM), c((size_t)XDIM * YDIM);
        make_frame(a, 11); make_frame(b, 12); make_frame(c, 13);
        std::vector<float> fl((size_t)XDIM * YDIM * 2), ref((size_t)XDIM * YDIM * 2);
        rc::Mat ma(YDIM, XDIM, 1, 1, a.data()), mb(YDIM, XDIM, 1, 1, b.data()), mc(YDIM, XDIM, 1, 1, c.data());
        rc::Mat mfl(YDIM, XDIM, 2, 4, fl.data()), mref(YDIM, XDIM, 2, 4, ref.data());
        REQUIRE(rcflow_stream_reset(pipe.context(), 0) == 0);
        REQUIRE(!pipe.pushFrame(ma, mfl, 0.5, 2, 3, 2, 15, 1.2, 0));          // primes
        REQUIRE(pipe.pushFrame(mb, mfl, 0.5, 2, 3, 2, 15, 1.2, 0));
        REQUIRE(pipe.pushFrame(mc, mfl, 0.5, 2, 3, 2, 15, 1.2, 0));
        pipe.calcOpticalFlowFarneback(mb, mc, mref, 0.5, 2, 3, 2, 15, 1.2, 0);
        REQUIRE(memcmp(fl.data(), ref.data(), fl.size() * sizeof(float)) == 0);
        REQUIRE(!pipe.pushFrame(ma, mfl, 0.5, 2, 10, 3, 15, 1.2, 256));      // other parameters: primes again
        // the same loop without the staging copy: the frame is produced into the slot's page-locked buffer
        REQUIRE(rcflow_stream_reset(pipe.context(), 0) == 0);
        const std::vector<uint8_t>* src[3] = {&a, &b, &c};
        for (int t = 0; t < 3; t++) {
            rc::Mat buf = pipe.frameBuffer();
            REQUIRE(buf.rows == YDIM && buf.cols == XDIM && buf.step >= (size_t)XDIM);
            for (int y = 0; y < YDIM; y++) memcpy((uint8_t*)buf.data + (size_t)y * buf.step, src[t]->data() + (size_t)y * XDIM, XDIM);
            REQUIRE(pipe.pushAcquired(mfl, 0.5, 2, 3, 2, 15, 1.2, 0) == (t > 0));
        }
        REQUIRE(memcmp(fl.data(), ref.data(), fl.size() * sizeof(float)) == 0);
        rc_farneback_params prm = {0.5, 2, 3, 2, 15, 1.2, 0};
        REQUIRE(rcflow_push_frame_acquired(pipe.context(), 0, &prm) == RC_ESTATE);       // nothing acquired

        // the whole iteration behind one call (Pipeline::loopStep = rcflow_frame_loop_step), eager and as one hipGraph
        // launch per frame: the wave mask after three flows equals the one the separate calls produce
        std::vector<uint8_t> want((size_t)XDIM * YDIM), got((size_t)XDIM * YDIM);
        rc::Mat mwant(YDIM, XDIM, 1, 1, want.data());
        REQUIRE(rcflow_analysis_reset(pipe.context(), 0, XDIM, YDIM) == 0);
        REQUIRE(rcflow_stream_reset(pipe.context(), 0) == 0);
        int hh[RC_HIST_BINS], hhs = 0, hh2[RC_HIST_DIRECTIONS][RC_HIST_BINS], hhs2[RC_HIST_DIRECTIONS];
        float hu, hu2[RC_HIST_DIRECTIONS], hp[RC_HIST_DIRECTIONS];
        rc::Mat none2;
        for (int t = 0, fc = 0; t < 4; t++) {
            rc::Mat m(YDIM, XDIM, 1, 1, (void*)src[t % 3]->data());
            if (!pipe.pushFrame(m, none2, 0.5, 2, 3, 2, 15, 1.2, 0)) continue;
            fc++;
            pipe.streamline_field(2, 1);
            pipe.create_histogram(hh, hhs, hh2, hhs2, hu, hu2, hp);
            pipe.create_flow_and_accumulationbuffer(mwant, fc);
        }
        for (int graph = 0; graph < 2; graph++) {
            REQUIRE(rcflow_analysis_reset(pipe.context(), 0, XDIM, YDIM) == 0);
            REQUIRE(rcflow_stream_reset(pipe.context(), 0) == 0);
            for (int t = 0; t < 4; t++) {
                rc::Mat buf = pipe.frameBuffer();
                for (int y = 0; y < YDIM; y++) memcpy((uint8_t*)buf.data + (size_t)y * buf.step, src[t % 3]->data() + (size_t)y * XDIM, XDIM);
                REQUIRE(pipe.loopStep(0.5, 2, 3, 2, 15, 1.2, 0, 2.f, 1, nullptr, 0, 100.f, 0.5f, 0.2f, graph == 1) == (t > 0));
            }
            REQUIRE(rcflow_sync(pipe.context(), 0) == 0);
            REQUIRE(hipMemcpy(got.data(), pipe.outmaskDevice(), got.size(), hipMemcpyDeviceToHost) == hipSuccess);
            REQUIRE(memcmp(got.data(), want.data(), got.size()) == 0);
        }
    }

    // the asynchronous host loop: frames from host memory through the page-locked double buffer, the flow field
    // never leaves the device, the analysis reads the resident field; and the reference's timing buckets
    {
        std::vector<uint8_t> fr[4];
        for (int t = 0; t < 4; t++) make_frame(fr[t], 20 + t);
        rc::Mat none;
        int h1[RC_HIST_BINS], hs1 = 0, h2d1[RC_HIST_DIRECTIONS][RC_HIST_BINS], hs2d1[RC_HIST_DIRECTIONS];
        int h2[RC_HIST_BINS], hs2 = 0, h2d2[RC_HIST_DIRECTIONS][RC_HIST_BINS], hs2d2[RC_HIST_DIRECTIONS];
        float U, U2[RC_HIST_DIRECTIONS], P[RC_HIST_DIRECTIONS];
        REQUIRE(rcflow_analysis_reset(pipe.context(), 0, XDIM, YDIM) == 0);
        REQUIRE(rcflow_stream_reset(pipe.context(), 0) == 0);
        REQUIRE(rcflow_profile_reset(pipe.context()) == 0 && rcflow_profile_enable(pipe.context(), 1) == 0);
        for (int t = 0; t < 4; t++) {
            rc::Mat m(YDIM, XDIM, 1, 1, fr[t].data());
            if (pipe.pushFrame(m, none, 0.5, 2, 3, 2, 15, 1.2, 0)) pipe.create_histogram(h1, hs1, h2d1, hs2d1, U, U2, P);
        }
        REQUIRE(rcflow_profile_enable(pipe.context(), 0) == 0);
        const char* names[RC_PROFILE_BUCKETS];
        double ms[RC_PROFILE_BUCKETS];
        REQUIRE(rcflow_profile_read_buckets(pipe.context(), names, ms) == RC_PROFILE_BUCKETS);
        REQUIRE(!strcmp(names[0], "farneback") && !strcmp(names[2], "threshold") && !strcmp(names[6], "stream"));
        REQUIRE(ms[0] > 0 && ms[2] > 0 && ms[1] == 0 && ms[5] == 0);
        printf("Time spent on farneback: %f ms\nTime spent on thresholds: %f ms\n", ms[0], ms[2]);   // ripcurrents.cpp:518-520
        REQUIRE(rcflow_analysis_reset(pipe.context(), 0, XDIM, YDIM) == 0);
        for (int t = 0; t < 3; t++) {
            rc::Mat a(YDIM, XDIM, 1, 1, fr[t].data()), b(YDIM, XDIM, 1, 1, fr[t + 1].data());
            pipe.calcOpticalFlowFarneback(a, b, none, 0.5, 2, 3, 2, 15, 1.2, 0);
            pipe.create_histogram(h2, hs2, h2d2, hs2d2, U, U2, P);
        }
        REQUIRE(hs1 == hs2 && hs1 > 0 && memcmp(h2d1, h2d2, sizeof(h2d1)) == 0);
        // the collective behind the C ABI: a world of one rank is the identity (no RCCL involved)
        REQUIRE(rcflow_comm_init(pipe.context(), nullptr, 0, 1) == 0);
        REQUIRE(rcflow_allreduce_hist(pipe.context(), 0, nullptr) == 0 && rcflow_allreduce_hist_join(pipe.context(), 0) == 0);
        int32_t* d_sum = nullptr;
        REQUIRE(rcflow_allreduce_hist_result(pipe.context(), &d_sum) == 0 && d_sum);
        REQUIRE(rcflow_thresholds_words_dev(pipe.context(), 0, d_sum) == 0 && rcflow_sync(pipe.context(), 0) == 0);
        std::vector<int32_t> words(RC_HIST_WORDS);
        REQUIRE(hipMemcpy(words.data(), d_sum, RC_HIST_WORDS * 4, hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(words[RC_HIST_BINS + RC_HIST_DIRECTIONS * RC_HIST_BINS] == hs2);
        REQUIRE(rcflow_comm_destroy(pipe.context()) == 0);
        REQUIRE(rcflow_allreduce_hist(pipe.context(), 0, nullptr) == RC_ECOMM);
    }

    bool threw = false;
    try {
        rc::Mat prev(YDIM, XDIM, 1, 1, f2.data()), bad;
        pipe.calcOpticalFlowFarneback(prev, prev, bad, 1.5, 2, 3, 2, 15, 1.2, 0);
    } catch (const rc::Error& e) { threw = e.code == RC_EINVAL; }
    REQUIRE(threw);
    printf("test_module: ok\n");
    return 0;
}
