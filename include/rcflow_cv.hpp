// rcflow_cv.hpp -- the literal drop-in for cv::calcOpticalFlowFarneback.
//
// Compiled only where OpenCV headers exist (they do not in this repository's build image;
// the reference pins OpenCV 4.1.0, RipCurrents_main/CMakeCache.txt:334).  tests/cpp/test_dropin.cpp compiles it
// against a minimal stand-in for <opencv2/core.hpp> (tests/cpp/opencv_standin: test scaffolding, not OpenCV) so that
// the adapter's code is at least built and run through the C ABI.  Signature-identical
// to the call at RipCurrents_main/ripcurrents.cpp:215 / main.cpp:264:
//
//     rc::calcOpticalFlowFarneback(prev, next, flow, 0.5, 2, 3, 2, 15, 1.2, 0);
//
// Like upstream it (re)allocates `flow` as CV_32FC2 of prev's size and throws cv::Exception
// on bad arguments.  UMat arguments are mapped with getMat(), i.e. the host-pointer entry point
// (copy in, compute, copy out); for a zero-copy path keep frames on the device and call
// rcflow_push_frame_dev / rcflow_push_clip_dev / rcflow_farneback_clip_dev directly.
#pragma once

#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>

#include <mutex>
#include <vector>

#include "rcflow.h"

namespace rc {

// cv::calcOpticalFlowFarneback is re-entrant (any thread, any size, no state between calls); an rc_ctx is one GPU
// context whose stream slot serves one call at a time.  Calls therefore lease a context from a process-wide pool: a
// free one that is large enough is reused, otherwise one is created (a free one that is too small is retired);
// concurrent callers each hold their own.  Contexts live until the process ends.
class ContextLease {
public:
    ContextLease(int w, int h) : w_(w), h_(h) {
        {
            std::lock_guard<std::mutex> g(pool().m);
            auto& fr = pool().free;
            for (size_t i = 0; i < fr.size(); i++)
                if (fr[i].w >= w && fr[i].h >= h) { e_ = fr[i]; fr.erase(fr.begin() + i); return; }
            if (!fr.empty()) {              // too small for this frame: retire one, its replacement covers both sizes
                Entry old = fr.back();
                fr.pop_back();
                w_ = old.w > w ? old.w : w;
                h_ = old.h > h ? old.h : h;
                rcflow_destroy(old.ctx);
            }
        }
        e_.w = w_; e_.h = h_;
        if (rcflow_create(&e_.ctx, 0, e_.w, e_.h, 1) != RC_OK) CV_Error(cv::Error::GpuApiCallError, rcflow_last_error());
    }
    ~ContextLease() {
        if (!e_.ctx) return;
        std::lock_guard<std::mutex> g(pool().m);
        pool().free.push_back(e_);
    }
    ContextLease(const ContextLease&) = delete;
    ContextLease& operator=(const ContextLease&) = delete;
    rc_ctx* get() const { return e_.ctx; }

private:
    struct Entry { rc_ctx* ctx = nullptr; int w = 0, h = 0; };
    struct Pool { std::mutex m; std::vector<Entry> free; };
    static Pool& pool() { static Pool* p = new Pool(); return *p; }     // never destroyed: outlives the HIP runtime's teardown
    Entry e_;
    int w_, h_;
};

inline void calcOpticalFlowFarneback(cv::InputArray _prev, cv::InputArray _next, cv::InputOutputArray _flow,
                                     double pyr_scale, int levels, int winsize, int iterations, int poly_n,
                                     double poly_sigma, int flags) {
    cv::Mat prev = _prev.getMat(), next = _next.getMat();
    CV_Assert(prev.size() == next.size() && prev.type() == CV_8UC1 && next.type() == CV_8UC1 && pyr_scale < 1);
    _flow.create(prev.size(), CV_32FC2);
    cv::Mat flow = _flow.getMat();
    ContextLease lease(prev.cols, prev.rows);
    int rc = rcflow_farneback_u8(lease.get(), 0, prev.data, prev.step, next.data,
                                 next.step, prev.cols, prev.rows, flow.ptr<float>(), flow.step, pyr_scale, levels,
                                 winsize, iterations, poly_n, poly_sigma, flags);
    if (rc != RC_OK) CV_Error(rc == RC_EINVAL ? cv::Error::StsBadArg : cv::Error::GpuApiCallError, rcflow_last_error());
}

// cv::calcOpticalFlowPyrLK with its own signature (Streakline.cpp:32, ripcurrents_module.cpp:716,
// :738, :775, :1162): 8UC1 images, vector<Point2f> points.
inline void calcOpticalFlowPyrLK(cv::InputArray _prev, cv::InputArray _next, cv::InputArray _prevPts,
                                 cv::InputOutputArray _nextPts, cv::OutputArray _status, cv::OutputArray _err,
                                 cv::Size winSize = cv::Size(21, 21), int maxLevel = 3,
                                 cv::TermCriteria criteria = cv::TermCriteria(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, 30, 0.01),
                                 int flags = 0, double minEigThreshold = 1e-4) {
    cv::Mat prev = _prev.getMat(), next = _next.getMat(), prevPts = _prevPts.getMat();
    CV_Assert(prev.size() == next.size() && prev.type() == CV_8UC1 && next.type() == CV_8UC1);
    const int n = prevPts.checkVector(2, CV_32F, true);
    CV_Assert(n >= 0);
    if (!(flags & cv::OPTFLOW_USE_INITIAL_FLOW)) _nextPts.create(prevPts.size(), prevPts.type(), -1, true);
    cv::Mat nextPts = _nextPts.getMat();
    CV_Assert(nextPts.checkVector(2, CV_32F, true) == n);
    _status.create(n, 1, CV_8U, -1, true);
    cv::Mat status = _status.getMat(), err;
    if (_err.needed()) { _err.create(n, 1, CV_32F, -1, true); err = _err.getMat(); }
    ContextLease lease(prev.cols, prev.rows);
    int rc = rcflow_pyrlk_u8(lease.get(), 0, prev.data, prev.step, next.data, next.step, prev.cols,
                             prev.rows, prevPts.ptr<float>(), nextPts.ptr<float>(), n, status.data,
                             err.empty() ? nullptr : err.ptr<float>(), winSize.width, winSize.height, maxLevel,
                             criteria.type, criteria.maxCount, criteria.epsilon, flags, minEigThreshold);
    if (rc != RC_OK) CV_Error(rc == RC_EINVAL ? cv::Error::StsBadArg : cv::Error::GpuApiCallError, rcflow_last_error());
}

}  // namespace rc
#endif  // OpenCV present
