// The literal drop-in for ripcurrents.cpp:215 / main.cpp:264, compiled and run: include/rcflow_cv.hpp built against the
// minimal <opencv2/core.hpp> stand-in of tests/cpp/opencv_standin (TEST SCAFFOLDING, not OpenCV: it pins nothing about
// parity with OpenCV; it lets this repository's own adapter be compiled where OpenCV does not exist) and driven with
// the reference's argument lists through the C ABI.  The CPU oracle is linked as checker.  Built by
// __graft_entry__.build(), run on the GPU box by tests/test_gpu_cpp_module.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/rcflow_cv.hpp"
#include "../../oracle/rc_oracle.h"

#define REQUIRE(c)                                                        \
    do {                                                                  \
        if (!(c)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } \
    } while (0)

static void make_frame(std::vector<uint8_t>& f, int w, int h, int t) {   // moving smooth texture
    f.resize((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double u = x - 1.25 * t, v = y + 0.75 * t;
            double s = 128 + 40 * std::sin(u / 7.0) * std::cos(v / 9.0) + 30 * std::sin((u + v) / 13.0) + 20 * std::cos(u / 3.1 - v / 4.3);
            f[(size_t)y * w + x] = (uint8_t)std::lrint(std::fmin(255.0, std::fmax(0.0, s)));
        }
}

// the fraction of pixels of `got` within 1e-3 px of the oracle's flow for the same call
static double agree(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b, int w, int h, const cv::Mat& got, int flags, bool* exact) {
    std::vector<float> ref((size_t)w * h * 2);
    if (orc_farneback_u8(a.data(), w, b.data(), w, w, h, ref.data(), (size_t)w * 8, 0.5, 2, 3, 2, 15, 1.2, flags, 2) != 0) return -1;
    size_t good = 0;
    bool same = true;
    for (int y = 0; y < h; y++) {
        const float* g = got.ptr<float>(y);
        for (int x = 0; x < 2 * w; x += 2) {
            const float* r = &ref[((size_t)y * w) * 2 + x];
            if (std::fabs(g[x] - r[0]) <= 1e-3f && std::fabs(g[x + 1] - r[1]) <= 1e-3f) good++;
            if (std::memcmp(g + x, r, 8) != 0) same = false;
        }
    }
    if (exact) *exact = same;
    return (double)good / ((size_t)w * h);
}

int main() {
    const int W = 320, H = 240;
    std::vector<uint8_t> f0, f1;
    make_frame(f0, W, H, 0);
    make_frame(f1, W, H, 1);
    cv::Mat prev(H, W, CV_8UC1, f0.data()), next(H, W, CV_8UC1, f1.data()), flow;        // flow: empty, the callee allocates it
    // ripcurrents.cpp:215, argument for argument
    rc::calcOpticalFlowFarneback(prev, next, flow, 0.5, 2, 3, 2, 15, 1.2, 0);
    REQUIRE(flow.rows == H && flow.cols == W && flow.type() == CV_32FC2 && flow.data);
    REQUIRE(agree(f0, f1, W, H, flow, 0, nullptr) > 0.995);
    // main.cpp:264 (the CMake target's call): the default path is upstream's operation order -> the oracle bit for bit
    cv::Mat flow_g;
    bool same = false;
    rc::calcOpticalFlowFarneback(prev, next, flow_g, 0.5, 2, 3, 2, 15, 1.2, cv::OPTFLOW_FARNEBACK_GAUSSIAN);
    REQUIRE(agree(f0, f1, W, H, flow_g, 256, &same) > 0.999 && same);
    // upstream's preconditions raise cv::Exception (the reference never catches: ripcurrents.cpp:215)
    {
        std::vector<uint8_t> small((size_t)100 * 80, 0);
        cv::Mat other(80, 100, CV_8UC1, small.data());
        bool threw = false;
        try { rc::calcOpticalFlowFarneback(prev, other, flow, 0.5, 2, 3, 2, 15, 1.2, 0); } catch (const cv::Exception&) { threw = true; }
        REQUIRE(threw);
        threw = false;
        try { rc::calcOpticalFlowFarneback(prev, next, flow, 1.0, 2, 3, 2, 15, 1.2, 0); } catch (const cv::Exception&) { threw = true; }
        REQUIRE(threw);
    }
    // re-entrant like cv::calcOpticalFlowFarneback: two threads, two frame sizes, at once -- every result equal to
    // the one the same call gives alone
    {
        const int W2 = 200, H2 = 152;
        std::vector<uint8_t> g0, g1;
        make_frame(g0, W2, H2, 3);
        make_frame(g1, W2, H2, 4);
        cv::Mat p2(H2, W2, CV_8UC1, g0.data()), n2(H2, W2, CV_8UC1, g1.data()), alone2;
        rc::calcOpticalFlowFarneback(p2, n2, alone2, 0.5, 2, 3, 2, 15, 1.2, 0);
        int bad[2] = {0, 0};
        auto worker = [&](int id) {
            for (int rep = 0; rep < 6; rep++) {
                cv::Mat out;
                if (id == 0) rc::calcOpticalFlowFarneback(prev, next, out, 0.5, 2, 3, 2, 15, 1.2, 0);
                else rc::calcOpticalFlowFarneback(p2, n2, out, 0.5, 2, 3, 2, 15, 1.2, 0);
                const cv::Mat& want = id == 0 ? flow : alone2;
                for (int y = 0; y < want.rows; y++)
                    if (std::memcmp(out.ptr<float>(y), want.ptr<float>(y), (size_t)want.cols * 8) != 0) bad[id]++;
            }
        };
        std::thread t0(worker, 0), t1(worker, 1);
        t0.join(); t1.join();
        REQUIRE(bad[0] == 0 && bad[1] == 0);
    }
    // Streakline.cpp:32's call through the adapter: vector<Point2f> in and out, status / err vectors created
    {
        std::vector<cv::Point2f> pts(12), moved;
        for (int i = 0; i < 12; i++) { pts[i].x = 30.f + 22.f * i; pts[i].y = 40.f + 13.f * i; }
        std::vector<uint8_t> status;
        std::vector<float> err;
        rc::calcOpticalFlowPyrLK(prev, next, pts, moved, status, err, cv::Size(50, 50), 3,
                                 cv::TermCriteria(cv::TermCriteria::COUNT | cv::TermCriteria::EPS, 30, 0.1), 10, 1e-4);
        REQUIRE(moved.size() == 12 && status.size() == 12 && err.size() == 12);
        std::vector<float> onext(24), oerr(12);
        std::vector<uint8_t> ostat(12);
        REQUIRE(orc_pyrlk(f0.data(), W, f1.data(), W, W, H, (const float*)pts.data(), onext.data(), 12, ostat.data(), oerr.data(), 50, 50, 3,
                          3, 30, 0.1, 10, 1e-4) == 0);
        for (int i = 0; i < 12; i++) {
            REQUIRE(status[i] == ostat[i]);
            if (status[i]) REQUIRE(std::fabs(moved[i].x - onext[2 * i]) < 2e-3f && std::fabs(moved[i].y - onext[2 * i + 1]) < 2e-3f);
        }
    }
    printf("test_dropin ok\n");
    return 0;
}
