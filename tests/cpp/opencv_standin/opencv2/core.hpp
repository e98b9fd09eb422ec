// TEST SCAFFOLDING, NOT OPENCV.  A minimal stand-in for the parts of <opencv2/core.hpp> that include/rcflow_cv.hpp
// touches, so that this repository's own adapter (rc::calcOpticalFlowFarneback / rc::calcOpticalFlowPyrLK) can be
// compiled and driven through the C ABI in an image that has no OpenCV (tests/cpp/test_dropin.cpp).  It holds no
// OpenCV code and no algorithm: a reference-counted-free Mat (pointer, byte step, type), the proxy-array classes with
// the handful of members the adapter calls, TermCriteria, Size, the error macros.  It pins NOTHING about parity with
// OpenCV and builds nothing of the reference; names and member semantics follow the public OpenCV API so that the
// adapter source is the same text a real OpenCV build would compile.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)
#define CV_MAT_DEPTH(t) ((t) & 7)
#define CV_MAT_CN(t) ((((t) >> CV_CN_SHIFT) & 511) + 1)

namespace cv {

namespace Error { enum Code { StsBadArg = -5, StsAssert = -215, GpuApiCallError = -217 }; }
enum { OPTFLOW_USE_INITIAL_FLOW = 4, OPTFLOW_LK_GET_MIN_EIGENVALS = 8, OPTFLOW_FARNEBACK_GAUSSIAN = 256 };

class Exception : public std::runtime_error {
public:
    Exception(int c, const std::string& m) : std::runtime_error(m), code(c) {}
    int code;
};

struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
};
struct Point2f { float x = 0, y = 0; };

struct TermCriteria {
    enum Type { COUNT = 1, MAX_ITER = COUNT, EPS = 2 };
    int type = 0, maxCount = 0;
    double epsilon = 0;
    TermCriteria() {}
    TermCriteria(int t, int n, double e) : type(t), maxCount(n), epsilon(e) {}
};

class Mat {
public:
    int rows = 0, cols = 0, flags_ = 0;
    uint8_t* data = nullptr;
    size_t step = 0;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* p, size_t s = 0) : rows(r), cols(c), flags_(type), data((uint8_t*)p), step(s ? s : (size_t)c * esz(type)) {}
    Mat(const Mat& o) = default;                 // shallow, like cv::Mat (this stand-in never frees: test lifetime only)
    Mat& operator=(const Mat& o) = default;
    static size_t esz(int type) { return (size_t)(CV_MAT_DEPTH(type) == CV_8U ? 1 : 4) * CV_MAT_CN(type); }
    void create(int r, int c, int type) {
        if (data && rows == r && cols == c && flags_ == type) return;
        rows = r; cols = c; flags_ = type; step = (size_t)c * esz(type);
        data = (uint8_t*)std::malloc(step * (size_t)(r > 0 ? r : 1));
    }
    void create(Size s, int type) { create(s.height, s.width, type); }
    Size size() const { return Size(cols, rows); }
    int type() const { return flags_; }
    bool empty() const { return !data || rows * cols == 0; }
    template <typename T> T* ptr(int r = 0) { return (T*)(data + step * (size_t)r); }
    template <typename T> const T* ptr(int r = 0) const { return (const T*)(data + step * (size_t)r); }
    // number of elements of `cn` channels when the matrix is an N x 1 / 1 x N / N x cn list of that depth, else -1
    int checkVector(int cn, int depth, bool /*requireContinuous*/) const {
        if (CV_MAT_DEPTH(flags_) != depth) return -1;
        const int ch = CV_MAT_CN(flags_);
        if (ch == cn && (rows == 1 || cols == 1)) return rows * cols;
        if (ch == 1 && cols == cn) return rows;
        return -1;
    }
};

// proxy arrays: a Mat or a std::vector<Point2f> / std::vector<uchar> / std::vector<float> behind one interface
class _InputArray {
public:
    _InputArray() {}
    _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
    _InputArray(const std::vector<Point2f>& v) : pts_(const_cast<std::vector<Point2f>*>(&v)) {}
    Mat getMat() const {
        if (m_) return *m_;
        if (pts_) return Mat((int)pts_->size(), 1, CV_32FC2, pts_->data());
        if (u8_) return Mat((int)u8_->size(), 1, CV_8UC1, u8_->data());
        if (f32_) return Mat((int)f32_->size(), 1, CV_32FC1, f32_->data());
        return Mat();
    }
protected:
    Mat* m_ = nullptr;
    std::vector<Point2f>* pts_ = nullptr;
    std::vector<uint8_t>* u8_ = nullptr;
    std::vector<float>* f32_ = nullptr;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat& m) { m_ = &m; }
    _OutputArray(std::vector<Point2f>& v) { pts_ = &v; }
    _OutputArray(std::vector<uint8_t>& v) { u8_ = &v; }
    _OutputArray(std::vector<float>& v) { f32_ = &v; }
    bool needed() const { return m_ || pts_ || u8_ || f32_; }
    void create(Size s, int type, int = -1, bool = false) const { create(s.height, s.width, type); }
    void create(int r, int c, int type, int = -1, bool = false) const {
        if (m_) m_->create(r, c, type);
        else if (pts_) pts_->resize((size_t)r * c);
        else if (u8_) u8_->resize((size_t)r * c);
        else if (f32_) f32_->resize((size_t)r * c);
    }
};
class _InputOutputArray : public _OutputArray {
public:
    using _OutputArray::_OutputArray;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _InputOutputArray& InputOutputArray;
inline const _OutputArray& noArray() { static _OutputArray a; return a; }

}  // namespace cv

#define CV_Error(code, msg) throw cv::Exception((int)(code), std::string(msg))
#define CV_Assert(expr) do { if (!(expr)) throw cv::Exception(cv::Error::StsAssert, #expr); } while (0)
