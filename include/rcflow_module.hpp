// rcflow_module.hpp -- C++ host-side mirror of the reference's interface for the hot path,
// over the C ABI of librcflow (include/rcflow.h).  Header-only; needs the HIP runtime API
// for host<->device copies (compile with hipcc, or g++ -D__HIP_PLATFORM_AMD__
// -I/opt/rocm/include ... -lamdhip64 -lrcflow).
//
// Names, argument order and meaning follow /root/reference/RipCurrents_main:
//   calcOpticalFlowFarneback   the cv:: call at ripcurrents.cpp:215, main.cpp:264,...
//   create_histogram           ripcurrents.hpp:39   ripcurrents_module.cpp:89-144
//   create_flow                ripcurrents.hpp:50   ripcurrents_module.cpp:153-182
//   create_accumulationbuffer  ripcurrents.hpp:52   ripcurrents_module.cpp:189-212
//   streamline_field           ripcurrents.hpp:22   ripcurrents_module.cpp:608-648
//   streamline                 ripcurrents.hpp:23   ripcurrents_module.cpp:486-528
//   Streakline                 Streakline.hpp:8-20  Streakline.cpp:11-71
//   Timeline, PopulationMap    ripcurrents.hpp:64-75, 86-95  ripcurrents_module.cpp:751-807, 1140-1196
// rc::Mat is a non-owning view with cv::Mat's fields (data, step, rows, cols); with OpenCV
// present, include/rcflow_cv.hpp converts cv::Mat to it.  Errors are thrown as
// rc::Error (the reference's OpenCV calls throw cv::Exception and are never caught).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "rcflow.h"

namespace rc {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc) {
    if (rc < 0) throw Error(rc, std::string("rcflow error ") + std::to_string(rc) + ": " + rcflow_last_error());
}
inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(RC_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

// cv::Mat-compatible view of host memory (interleaved channels, byte step).
struct Mat {
    void* data = nullptr;
    size_t step = 0;
    int rows = 0, cols = 0;
    int channels = 1, elem = 1;   // elem = bytes per channel
    Mat() = default;
    Mat(int r, int c, int ch, int el, void* d, size_t st = 0)
        : data(d), step(st ? st : (size_t)c * ch * el), rows(r), cols(c), channels(ch), elem(el) {}
    size_t row_bytes() const { return (size_t)cols * channels * elem; }
    bool empty() const { return !data || rows <= 0 || cols <= 0; }
};

typedef struct { float x, y; } Pixel2;   // cv::Point_<float> (ripcurrents.hpp:19)

// One stream slot of one GPU context plus the device copies the module functions work on.
// The flow field stays resident on the device between calcOpticalFlowFarneback and the
// analysis calls, like `current` does on the host in ripcurrents.cpp:221-440.
class Pipeline {
  public:
    Pipeline(int xdim, int ydim, int device = 0) : w_(xdim), h_(ydim) {
        check(rcflow_create(&ctx_, device, xdim, ydim, 1));
        try {
            hip_check(hipMalloc(&d_frames_, (size_t)2 * w_ * h_), "hipMalloc frames");
            hip_check(hipMalloc(&d_flow_, (size_t)w_ * h_ * 8), "hipMalloc flow");
            hip_check(hipMalloc(&d_mask_, (size_t)w_ * h_), "hipMalloc mask");
            flow_src_ = (const float*)d_flow_;
            check(rcflow_analysis_reset(ctx_, 0, w_, h_));
        } catch (...) {
            release();
            throw;
        }
    }
    ~Pipeline() { release(); }
    Pipeline(const Pipeline&) = delete;
    Pipeline& operator=(const Pipeline&) = delete;

    rc_ctx* context() { return ctx_; }
    const float* device_flow() const { return flow_src_; }

    // cv::calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations,
    // poly_n, poly_sigma, flags): 8UC1 in, CV_32FC2 out (flow.data may be null: the field then
    // only stays resident for the analysis calls).
    // The frame loop itself (ripcurrents.cpp:194-221: capture, calcOpticalFlowFarneback(u_f2, u_f1, ...),
    // u_f1.copyTo(u_f2)) with the previous frame kept on the device: one upload and one pyramid + expansion
    // per frame (rcflow_push_frame_dev).  Returns false for the call that primes the stream (the first one, or
    // the first with other parameters): no flow yet.  The flow equals calcOpticalFlowFarneback(previous, frame).
    // The upload goes through the library's page-locked double buffer (rcflow_push_frame_u8) and is
    // asynchronous: with a null `flow` view the call returns while upload and kernels are still running, and the
    // analysis calls below queue behind them on the same stream; only a non-null `flow` waits and downloads.
    bool pushFrame(const Mat& frame, Mat& flow, double pyr_scale, int levels, int winsize, int iterations, int poly_n,
                   double poly_sigma, int flags) {
        if (frame.empty() || frame.channels != 1 || frame.elem != 1 || frame.cols != w_ || frame.rows != h_)
            throw Error(RC_EINVAL, "pushFrame: frame must be 8UC1 of the pipeline's size");
        if (flow.data && (flow.rows != h_ || flow.cols != w_ || flow.channels != 2 || flow.elem != 4))
            throw Error(RC_EINVAL, "flow must be CV_32FC2 of the frame size");
        rc_farneback_params p = {pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags};
        const int rc = rcflow_push_frame_u8(ctx_, 0, (const uint8_t*)frame.data, frame.step, w_, h_, &p);
        check(rc);
        if (rc == 1) return false;
        float* d = nullptr;
        check(rcflow_stream_flow_ptr(ctx_, 0, &d, nullptr, nullptr));
        flow_src_ = d;                                  // the analysis calls read the stream's resident field
        if (flow.data) check(rcflow_stream_flow_read(ctx_, 0, (float*)flow.data, flow.step));
        return true;
    }

    // The same loop without the staging copy: frameBuffer() is the slot's next page-locked staging buffer as an 8UC1
    // view of the pipeline's size -- make it the destination of the cvtColor at ripcurrents.cpp:210 -- and pushAcquired()
    // pushes it (rcflow_frame_buffer_acquire / rcflow_push_frame_acquired).
    Mat frameBuffer() {
        uint8_t* p = nullptr;
        size_t step = 0;
        check(rcflow_frame_buffer_acquire(ctx_, 0, w_, h_, &p, &step));
        return Mat(h_, w_, 1, 1, p, step);
    }
    bool pushAcquired(Mat& flow, double pyr_scale, int levels, int winsize, int iterations, int poly_n, double poly_sigma,
                      int flags) {
        if (flow.data && (flow.rows != h_ || flow.cols != w_ || flow.channels != 2 || flow.elem != 4))
            throw Error(RC_EINVAL, "flow must be CV_32FC2 of the frame size");
        rc_farneback_params p = {pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags};
        const int rc = rcflow_push_frame_acquired(ctx_, 0, &p);
        check(rc);
        if (rc == 1) return false;
        float* d = nullptr;
        check(rcflow_stream_flow_ptr(ctx_, 0, &d, nullptr, nullptr));
        flow_src_ = d;
        if (flow.data) check(rcflow_stream_flow_read(ctx_, 0, (float*)flow.data, flow.step));
        return true;
    }

    // One whole iteration of the frame loop (ripcurrents.cpp:194-479) behind one call: the frame produced into
    // frameBuffer() is pushed, then streamline_field, the seed streamlines (d_seeds: device, n x (x, y), advanced in
    // place; may be null), histogram + thresholds, create_flow + create_accumulationbuffer (the frame counter lives on the
    // device) and the mask's edges run on the resident field (rcflow_frame_loop_step).  Returns false for the call that
    // primes the stream.  The masks stay on the device: outmaskDevice() / edgesDevice(), or read them back as needed.
    bool loopStep(double pyr_scale, int levels, int winsize, int iterations, int poly_n, double poly_sigma, int flags,
                  float dt = 2.f, int streamline_iterations = 1, float* d_seeds = nullptr, int nseeds = 0, float seed_upper = 100.f,
                  float MID = 0.5f, float LOWER = 0.2f, bool use_graph = false) {
        if (!d_edges_) hip_check(hipMalloc(&d_edges_, (size_t)w_ * h_), "hipMalloc edges");
        rc_farneback_params p = {pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags};
        rc_frame_loop L;
        std::memset(&L, 0, sizeof(L));
        L.dt = dt; L.iterations = streamline_iterations;
        L.d_seeds = d_seeds; L.nseeds = nseeds; L.seed_variant = 3; L.seed_dt = dt; L.seed_iterations = streamline_iterations; L.seed_upper = seed_upper;
        L.MID = MID; L.LOWER = LOWER;
        L.d_outmask = (uint8_t*)d_mask_; L.mask_step = (size_t)w_; L.d_edges = (uint8_t*)d_edges_; L.edges_step = (size_t)w_;
        L.use_graph = use_graph ? 1 : 0;
        const int rc = rcflow_frame_loop_step(ctx_, 0, &p, &L);
        check(rc);
        if (rc == 1) return false;
        float* d = nullptr;
        check(rcflow_stream_flow_ptr(ctx_, 0, &d, nullptr, nullptr));
        flow_src_ = d;
        return true;
    }
    const uint8_t* outmaskDevice() const { return (const uint8_t*)d_mask_; }
    const uint8_t* edgesDevice() const { return (const uint8_t*)d_edges_; }

    void calcOpticalFlowFarneback(const Mat& prev, const Mat& next, Mat& flow, double pyr_scale, int levels,
                                  int winsize, int iterations, int poly_n, double poly_sigma, int flags) {
        if (prev.empty() || next.empty() || prev.rows != next.rows || prev.cols != next.cols ||
            prev.channels != 1 || prev.elem != 1 || prev.cols != w_ || prev.rows != h_)
            throw Error(RC_EINVAL, "calcOpticalFlowFarneback: prev/next must be 8UC1 of the pipeline's size");
        uint8_t* df = (uint8_t*)d_frames_;
        hip_check(hipMemcpy2D(df, w_, prev.data, prev.step, w_, h_, hipMemcpyHostToDevice), "upload prev");
        hip_check(hipMemcpy2D(df + (size_t)w_ * h_, w_, next.data, next.step, w_, h_, hipMemcpyHostToDevice), "upload next");
        rc_farneback_params p = {pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags};
        check(rcflow_farneback_dev(ctx_, 0, df, w_, df + (size_t)w_ * h_, w_, w_, h_, (float*)d_flow_, (size_t)w_ * 8, &p));
        flow_src_ = (float*)d_flow_;
        check(rcflow_sync(ctx_, 0));
        if (flow.data) {
            if (flow.rows != h_ || flow.cols != w_ || flow.channels != 2 || flow.elem != 4)
                throw Error(RC_EINVAL, "flow must be CV_32FC2 of the frame size");
            hip_check(hipMemcpy2D(flow.data, flow.step, d_flow_, (size_t)w_ * 8, (size_t)w_ * 8, h_, hipMemcpyDeviceToHost), "download flow");
        }
    }

    // cv::calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts, status, err, winSize, maxLevel,
    // criteria(type, maxCount, epsilon), flags, minEigThreshold) on 8UC1 frames -- Streakline.cpp:32,
    // ripcurrents_module.cpp:716, :738, :775, :1162.  criteria type: 1 = COUNT, 2 = EPS.
    void calcOpticalFlowPyrLK(const Mat& prev, const Mat& next, const std::vector<Pixel2>& prevPts,
                              std::vector<Pixel2>& nextPts, std::vector<unsigned char>& status, std::vector<float>& err,
                              int win_w = 21, int win_h = 21, int maxLevel = 3, int crit_type = 3, int maxCount = 30,
                              double epsilon = 0.01, int flags = 0, double minEigThreshold = 1e-4) {
        if (prev.empty() || next.empty() || prev.rows != next.rows || prev.cols != next.cols ||
            prev.channels != 1 || prev.elem != 1 || prev.cols != w_ || prev.rows != h_)
            throw Error(RC_EINVAL, "calcOpticalFlowPyrLK: prev/next must be 8UC1 of the pipeline's size");
        const int n = (int)prevPts.size();
        const bool initial = (flags & 4) != 0;
        if (initial && (int)nextPts.size() != n) throw Error(RC_EINVAL, "OPTFLOW_USE_INITIAL_FLOW needs nextPts of prevPts' size");
        nextPts.resize(n); status.assign(n, 0); err.assign(n, 0.f);
        if (n == 0) return;
        uint8_t* df = (uint8_t*)d_frames_;
        hip_check(hipMemcpy2D(df, w_, prev.data, prev.step, w_, h_, hipMemcpyHostToDevice), "upload prev");
        hip_check(hipMemcpy2D(df + (size_t)w_ * h_, w_, next.data, next.step, w_, h_, hipMemcpyHostToDevice), "upload next");
        // prevPts | nextPts | err | status in the pipeline's grow-only scratch buffer (no allocation in the frame loop)
        const size_t off_n = (size_t)n * 8, off_e = 2 * off_n, off_s = off_e + (size_t)n * 4;
        char* b = (char*)scratch(off_s + n);
        hipError_t e = hipMemcpy(b, prevPts.data(), off_n, hipMemcpyHostToDevice);
        if (e == hipSuccess && initial) e = hipMemcpy(b + off_n, nextPts.data(), off_n, hipMemcpyHostToDevice);
        int rc = e == hipSuccess ? rcflow_pyrlk_dev(ctx_, 0, df, w_, df + (size_t)w_ * h_, w_, w_, h_, (const float*)b,
                                                    (float*)(b + off_n), n, (uint8_t*)(b + off_s), (float*)(b + off_e), win_w,
                                                    win_h, maxLevel, crit_type, maxCount, epsilon, flags, minEigThreshold)
                                 : RC_EHIP;
        if (rc == RC_OK) rc = rcflow_sync(ctx_, 0);
        if (rc == RC_OK) {
            e = hipMemcpy(nextPts.data(), b + off_n, off_n, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(err.data(), b + off_e, (size_t)n * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(status.data(), b + off_s, n, hipMemcpyDeviceToHost);
        }
        check(rc);
        hip_check(e, "PyrLK point transfer");
    }

    // Replace the resident flow field (e.g. after host-side edits of `current`).
    void upload_flow(const Mat& current) {
        if (current.rows != h_ || current.cols != w_ || current.channels != 2 || current.elem != 4)
            throw Error(RC_EINVAL, "current must be CV_32FC2 of the frame size");
        hip_check(hipMemcpy2D(d_flow_, (size_t)w_ * 8, current.data, current.step, (size_t)w_ * 8, h_, hipMemcpyHostToDevice), "upload flow");
        flow_src_ = (float*)d_flow_;
    }

    // create_histogram(current, hist, histsum, hist2d, histsum2d, UPPER, UPPER2d, prop_above_upper)
    // on the resident flow field.  The counters are cumulative in the slot (never reset by the
    // reference); the arrays receive their current values.
    void create_histogram(int hist[RC_HIST_BINS], int& histsum, int hist2d[RC_HIST_DIRECTIONS][RC_HIST_BINS],
                          int histsum2d[RC_HIST_DIRECTIONS], float& UPPER, float UPPER2d[RC_HIST_DIRECTIONS],
                          float prop_above_upper[RC_HIST_DIRECTIONS]) {
        check(rcflow_histogram_dev(ctx_, 0, flow_src_, (size_t)w_ * 8, w_, h_));
        check(rcflow_thresholds_dev(ctx_, 0));
        int32_t hs = 0;
        check(rcflow_histogram_read(ctx_, 0, hist, &hist2d[0][0], &hs, histsum2d, &UPPER, UPPER2d, prop_above_upper));
        histsum = hs;
    }

    // create_flow + create_accumulationbuffer in one pass; outmask (8UC1, may be null view)
    // receives the wave mask.  UPPER / UPPER2d are the slot's (from create_histogram).
    void create_flow_and_accumulationbuffer(Mat& outmask, int framecount, float MID = 0.5f, float LOWER = 0.2f) {
        check(rcflow_classify_accumulate_dev(ctx_, 0, flow_src_, (size_t)w_ * 8, w_, h_, framecount, MID,
                                             LOWER, nullptr, 0, nullptr, 0, nullptr, 0, (uint8_t*)d_mask_, w_));
        check(rcflow_sync(ctx_, 0));
        if (outmask.data)
            hip_check(hipMemcpy2D(outmask.data, outmask.step, d_mask_, w_, w_, h_, hipMemcpyDeviceToHost), "download mask");
    }
    void accumulator(float* acc_x) { check(rcflow_accumulator_read(ctx_, 0, acc_x)); }

    // streamlines_mat.forEach(streamline_field(&pixel, distance, x, y, current, dt, iterations,
    // UPPER, prop_above_upper)) -- ripcurrents.cpp:229-231.  UPPER < 0: the slot's UPPER.
    void streamline_field(float dt, int iterations, float UPPER = -1.f) {
        check(rcflow_advect_field_dev(ctx_, 0, flow_src_, (size_t)w_ * 8, w_, h_, dt, iterations, UPPER));
    }
    void streamline_field_state(Pixel2* streamlines_mat, float* streamlines_distance) {
        check(rcflow_advect_field_read(ctx_, 0, (float*)streamlines_mat, streamlines_distance));
    }

    // streamline_displacement / streamline_total_motion / streamline_ratio
    // (ripcurrents_module.cpp:13-40, ripcurrents.cpp:233-257) on the resident streamline field:
    // streamoverlay_color must be 8UC3 of the frame size; only that image crosses PCIe.
    void streamline_displacement(Mat& streamoverlay_color) { display(0, streamoverlay_color); }
    void streamline_total_motion(Mat& streamoverlay_color) { display(1, streamoverlay_color); }
    void streamline_ratio(Mat& streamoverlay_color) { display(2, streamoverlay_color); }

    // for (s...) streamline(streampt + s, color, current, overlay, dt, iterations, UPPER, ...):
    // advances the seeds; `trace` (optional, n*iterations points) is what the host draws with
    // cv::line.  variant: see rcflow_advect_points_dev.
    void streamline(Pixel2* streampt, int n, float dt, int iterations, float UPPER, int variant = 0,
                    std::vector<Pixel2>* trace = nullptr) {
        if (n <= 0) return;
        const int iters = variant == 2 ? 100 : iterations;
        const size_t pts_bytes = ((size_t)n * 8 + 255) & ~(size_t)255;
        char* b = (char*)scratch(pts_bytes + (trace ? (size_t)n * iters * 8 : 0));     // grow-only, no per-frame allocation
        void *d_pts = b, *d_tr = trace ? b + pts_bytes : nullptr;
        hip_check(hipMemcpy(d_pts, streampt, (size_t)n * 8, hipMemcpyHostToDevice), "upload seeds");
        int rc = rcflow_advect_points_dev(ctx_, 0, (float*)d_pts, n, flow_src_, (size_t)w_ * 8, w_, h_, dt,
                                          iterations, UPPER, variant, (float*)d_tr);
        if (rc == RC_OK) rc = rcflow_sync(ctx_, 0);
        if (rc == RC_OK) {
            (void)hipMemcpy(streampt, d_pts, (size_t)n * 8, hipMemcpyDeviceToHost);
            if (trace) {
                trace->resize((size_t)n * iters);
                (void)hipMemcpy(trace->data(), d_tr, (size_t)n * iters * 8, hipMemcpyDeviceToHost);
            }
        }
        check(rc);
    }

    int width() const { return w_; }
    int height() const { return h_; }

  private:
    // grow-only device scratch for the per-frame helpers (seeds, traces, display image, LK points)
    void* scratch(size_t bytes) {
        if (bytes > scratch_bytes_) {
            check(rcflow_sync(ctx_, 0));
            if (d_scratch_) (void)hipFree(d_scratch_);
            d_scratch_ = nullptr; scratch_bytes_ = 0;
            const size_t want = bytes + bytes / 2;
            hip_check(hipMalloc(&d_scratch_, want), "hipMalloc scratch");
            scratch_bytes_ = want;
        }
        return d_scratch_;
    }
    void release() {
        if (d_scratch_) (void)hipFree(d_scratch_);
        d_scratch_ = nullptr; scratch_bytes_ = 0;
        if (d_frames_) (void)hipFree(d_frames_);
        if (d_flow_) (void)hipFree(d_flow_);
        if (d_mask_) (void)hipFree(d_mask_);
        if (d_edges_) (void)hipFree(d_edges_);
        d_frames_ = d_flow_ = d_mask_ = d_edges_ = nullptr;
        if (ctx_) rcflow_destroy(ctx_);
        ctx_ = nullptr;
    }
    void display(int which, Mat& bgr) {
        if (bgr.rows != h_ || bgr.cols != w_ || bgr.channels != 3 || bgr.elem != 1)
            throw Error(RC_EINVAL, "streamoverlay_color must be 8UC3 of the frame size");
        void* d = scratch((size_t)w_ * h_ * 3);
        int rc = rcflow_streamline_display_dev(ctx_, 0, which, (uint8_t*)d, (size_t)w_ * 3, nullptr);
        if (rc == RC_OK) rc = rcflow_sync(ctx_, 0);
        hipError_t e = rc == RC_OK ? hipMemcpy2D(bgr.data, bgr.step, d, (size_t)w_ * 3, (size_t)w_ * 3, h_, hipMemcpyDeviceToHost)
                                   : hipSuccess;
        check(rc);
        hip_check(e, "download display image");
    }
    rc_ctx* ctx_ = nullptr;
    int w_, h_;
    void *d_frames_ = nullptr, *d_flow_ = nullptr, *d_mask_ = nullptr, *d_edges_ = nullptr, *d_scratch_ = nullptr;
    size_t scratch_bytes_ = 0;
    const float* flow_src_ = nullptr;     // the field the analysis calls read: d_flow_ or the stream's resident field
};

// Streakline.hpp:8-20.  run(): runLK's bookkeeping (Streakline.cpp:22-71) with the vertices moved
// through the dense flow field resident in the pipeline (the main.cpp:961-977 precedent);
// runLK(): the same with the reference's own mover, sparse PyrLK.  Drawing stays with the caller.
class Streakline {
  public:
    int numberOfVertices;
    Pixel2 generationPoint;
    std::vector<Pixel2> vertices;
    int frameCount;

    explicit Streakline(Pixel2 pixel) : numberOfVertices(1), generationPoint(pixel), frameCount(1) {
        vertices.push_back(pixel);
    }

    void run(Pipeline& pipe, float dt = 1.f) {
        std::vector<Pixel2> next = vertices;
        pipe.streamline(next.data(), (int)next.size(), dt, 1, 0.f, /*variant=*/4);
        for (size_t i = 0; i < next.size(); i++)   // eliminate any large movement (Streakline.cpp:35-40)
            if (std::fabs(vertices[i].x - next[i].x) > pipe.width() * 0.1 ||
                std::fabs(vertices[i].y - next[i].y) > pipe.height() * 0.1)
                next[i] = vertices[i];
        vertices = next;
        vertices.insert(vertices.begin(), generationPoint);   // frameCount % 1 == 0 (Streakline.cpp:46-48)
        numberOfVertices = (int)vertices.size();
        frameCount++;
    }

    // Streakline::runLK(u_prev, u_current, outImg) as the reference runs it: the vertices are
    // moved by PyrLK (50x50, maxLevel 3, COUNT+EPS 30 / 0.1, flags 10, 1e-4; Streakline.cpp:32).
    void runLK(Pipeline& pipe, const Mat& u_prev, const Mat& u_current) {
        std::vector<Pixel2> next;
        std::vector<unsigned char> status;
        std::vector<float> err;
        pipe.calcOpticalFlowPyrLK(u_prev, u_current, vertices, next, status, err, 50, 50, 3, 3, 30, 0.1, 10, 1e-4);
        for (size_t i = 0; i < next.size(); i++)   // eliminate any large movement (Streakline.cpp:35-40)
            if (std::fabs(vertices[i].x - next[i].x) > pipe.width() * 0.1 ||
                std::fabs(vertices[i].y - next[i].y) > pipe.height() * 0.1)
                next[i] = vertices[i];
        vertices = next;
        vertices.insert(vertices.begin(), generationPoint);
        numberOfVertices = (int)vertices.size();
        frameCount++;
    }
};

// Timeline (ripcurrents.hpp:64-75, ripcurrents_module.cpp:751-807) and PopulationMap
// (ripcurrents.hpp:86-95, ripcurrents_module.cpp:1140-1196): point sets moved by sparse PyrLK with the
// parameters of :775 / :1162; every vertex takes its tracked position.  Drawing stays with the caller.
class Timeline {
  public:
    std::vector<Pixel2> vertices;
    Timeline(Pixel2 lineStart, Pixel2 lineEnd, int numberOfVertices) {
        float diffX = (lineEnd.x - lineStart.x) / numberOfVertices, diffY = (lineEnd.y - lineStart.y) / numberOfVertices;
        for (int i = 0; i <= numberOfVertices; i++) vertices.push_back(Pixel2{lineStart.x + diffX * i, lineStart.y + diffY * i});
    }
    void runLK(Pipeline& pipe, const Mat& u_prev, const Mat& u_current) {
        std::vector<Pixel2> next;
        std::vector<unsigned char> status;
        std::vector<float> err;
        pipe.calcOpticalFlowPyrLK(u_prev, u_current, vertices, next, status, err, 50, 50, 3, 3, 30, 0.1, 10, 1e-4);
        vertices = next;
    }
};

class PopulationMap {
  public:
    std::vector<Pixel2> vertices;
    // `unit_random` returns u in [0, 1] (the reference calls sranddev(); rand() / RAND_MAX); the
    // reference's formula start + (end - start) * (u + 1) is kept as written.
    template <class Rng>
    PopulationMap(Pixel2 rectStart, Pixel2 rectEnd, int numberOfVertices, Rng&& unit_random) {
        for (int i = 0; i < numberOfVertices; i++) {
            float randX = (float)((rectEnd.x - rectStart.x) * ((double)unit_random() + 1) + rectStart.x);
            float randY = (float)((rectEnd.y - rectStart.y) * ((double)unit_random() + 1) + rectStart.y);
            vertices.push_back(Pixel2{randX, randY});
        }
    }
    void runLK(Pipeline& pipe, const Mat& u_prev, const Mat& u_current) {
        std::vector<Pixel2> next;
        std::vector<unsigned char> status;
        std::vector<float> err;
        pipe.calcOpticalFlowPyrLK(u_prev, u_current, vertices, next, status, err, 50, 50, 3, 3, 30, 0.1, 10, 1e-4);
        vertices = next;
    }
};

}  // namespace rc
