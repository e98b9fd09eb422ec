// rcflow_cv.hpp -- the literal drop-in for cv::calcOpticalFlowFarneback.
//
// Compiled only where OpenCV headers exist (they do not in this repository's build image;
// the reference pins OpenCV 4.1.0, RipCurrents_main/CMakeCache.txt:334).  Signature-identical
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

#include "rcflow.h"

namespace rc {

inline rc_ctx* default_context(int w, int h) {
    static rc_ctx* ctx = nullptr;
    static int cw = 0, ch = 0;
    if (!ctx || w > cw || h > ch) {
        if (ctx) rcflow_destroy(ctx);
        ctx = nullptr;
        cw = w > cw ? w : cw;
        ch = h > ch ? h : ch;
        if (rcflow_create(&ctx, 0, cw, ch, 1) != RC_OK)
            CV_Error(cv::Error::GpuApiCallError, rcflow_last_error());
    }
    return ctx;
}

inline void calcOpticalFlowFarneback(cv::InputArray _prev, cv::InputArray _next, cv::InputOutputArray _flow,
                                     double pyr_scale, int levels, int winsize, int iterations, int poly_n,
                                     double poly_sigma, int flags) {
    cv::Mat prev = _prev.getMat(), next = _next.getMat();
    CV_Assert(prev.size() == next.size() && prev.type() == CV_8UC1 && next.type() == CV_8UC1 && pyr_scale < 1);
    _flow.create(prev.size(), CV_32FC2);
    cv::Mat flow = _flow.getMat();
    int rc = rcflow_farneback_u8(default_context(prev.cols, prev.rows), 0, prev.data, prev.step, next.data,
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
    int rc = rcflow_pyrlk_u8(default_context(prev.cols, prev.rows), 0, prev.data, prev.step, next.data, next.step, prev.cols,
                             prev.rows, prevPts.ptr<float>(), nextPts.ptr<float>(), n, status.data,
                             err.empty() ? nullptr : err.ptr<float>(), winSize.width, winSize.height, maxLevel,
                             criteria.type, criteria.maxCount, criteria.epsilon, flags, minEigThreshold);
    if (rc != RC_OK) CV_Error(rc == RC_EINVAL ? cv::Error::StsBadArg : cv::Error::GpuApiCallError, rcflow_last_error());
}

}  // namespace rc
#endif  // OpenCV present
