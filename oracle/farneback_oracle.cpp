// farneback_oracle.cpp -- CPU restatement of cv::calcOpticalFlowFarneback (A1-A7).
//
// TEST INFRASTRUCTURE, NOT PRODUCT.  PARITY UNPINNED (see rc_oracle.h).
//
// The reference calls cv::calcOpticalFlowFarneback at
//   RipCurrents_main/ripcurrents.cpp:215, main.cpp:264,609,742,961,1119,1481,
//   ripcurrents_module.cpp:712, main_old.cpp:324
// and the arithmetic is OpenCV's (pinned 4.1.0 by RipCurrents_main/CMakeCache.txt:334),
// which is not vendored.  This file restates the published CPU algorithm of
//   modules/video/src/optflow.cpp   FarnebackPrepareGaussian / FarnebackPolyExp /
//                                   FarnebackUpdateMatrices / FarnebackUpdateFlow_Blur /
//                                   FarnebackUpdateFlow_GaussianBlur /
//                                   FarnebackOpticalFlowImpl::calc
//   modules/imgproc/src/smooth.cpp  getGaussianKernel, GaussianBlur (sepFilter2D path)
//   modules/imgproc/src/resize.cpp  INTER_LINEAR for CV_32F
// keeping its float/double mix, border rules and round-half-even cvRound.
// Built with -ffp-contract=off so float expressions round like the scalar C++ there.

#include "rc_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

namespace {

inline int cv_round(double v) { return (int)std::nearbyint(v); }  // half-to-even (SSE2 cvtsd2si)
inline int cv_floor(double v) { return (int)std::floor(v); }

// BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

void parallel_rows(int rows, int nthreads, const std::function<void(int, int)>& fn) {
    if (nthreads <= 1 || rows < 2 * nthreads) {
        fn(0, rows);
        return;
    }
    std::vector<std::thread> pool;
    int chunk = (rows + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        int a = t * chunk, b = std::min(rows, a + chunk);
        if (a >= b) break;
        pool.emplace_back(fn, a, b);
    }
    for (auto& th : pool) th.join();
}

// ---------------------------------------------------------------- smooth.cpp
// cv::getGaussianKernel(n, sigma, CV_32F)
void gaussian_kernel(int n, double sigma, float* cf) {
    static const float small_tab[][7] = {
        {1.f},
        {0.25f, 0.5f, 0.25f},
        {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
        {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float* fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? small_tab[n >> 1] : nullptr;
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = fixed ? (double)fixed[i] : std::exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
}

// GaussianBlur(src32f, dst, Size(k,k), sigma, sigma, BORDER_DEFAULT) -> sepFilter2D:
// row pass (float buffer) then symmetric column pass.
void gaussian_blur_f32(const float* src, int w, int h, int ksize, double sigma, float* dst,
                       int nthreads) {
    if (ksize == 1) {
        std::memcpy(dst, src, sizeof(float) * (size_t)w * h);
        return;
    }
    std::vector<float> k(ksize);
    gaussian_kernel(ksize, std::max(sigma, 0.), k.data());
    const int r = ksize / 2;
    std::vector<float> tmp((size_t)w * h);
    parallel_rows(h, nthreads, [&](int ya, int yb) {
        std::vector<float> line(w + 2 * r);
        for (int y = ya; y < yb; y++) {
            const float* s = src + (size_t)y * w;
            for (int x = -r; x < w + r; x++) line[x + r] = s[reflect101(x, w)];
            float* d = tmp.data() + (size_t)y * w;
            if (ksize <= 5) {  // SymmRowSmallFilter: centre first, then mirrored pairs
                for (int x = 0; x < w; x++) {
                    const float* S = line.data() + x + r;
                    float s0 = S[0] * k[r];
                    if (ksize == 3) s0 = S[0] * k[r] + (S[-1] + S[1]) * k[r + 1];
                    else s0 = S[0] * k[r] + (S[-1] + S[1]) * k[r + 1] + (S[-2] + S[2]) * k[r + 2];
                    d[x] = s0;
                }
            } else {  // RowFilter: left-to-right accumulation
                for (int x = 0; x < w; x++) {
                    const float* S = line.data() + x;
                    float s0 = k[0] * S[0];
                    for (int j = 1; j < ksize; j++) s0 += k[j] * S[j];
                    d[x] = s0;
                }
            }
        }
    });
    parallel_rows(h, nthreads, [&](int ya, int yb) {
        std::vector<const float*> rows(ksize);
        for (int y = ya; y < yb; y++) {
            for (int j = -r; j <= r; j++)
                rows[j + r] = tmp.data() + (size_t)reflect101(y + j, h) * w;
            float* d = dst + (size_t)y * w;
            for (int x = 0; x < w; x++) {  // SymmColumnFilter
                float s0 = k[r] * rows[r][x];
                for (int j = 1; j <= r; j++) s0 += k[r + j] * (rows[r + j][x] + rows[r - j][x]);
                d[x] = s0;
            }
        }
    });
}

// ---------------------------------------------------------------- resize.cpp
// cv::resize(src, dst, Size(dw,dh), 0, 0, INTER_LINEAR) for CV_32FC(cn)
void resize_linear_f32(const float* src, int sw, int sh, int cn, float* dst, int dw, int dh,
                       int nthreads) {
    if (sw == dw && sh == dh) {
        std::memcpy(dst, src, sizeof(float) * (size_t)sw * sh * cn);
        return;
    }
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<float> alpha(2 * dw), beta(2 * dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = 1.f - fx;
        alpha[2 * dx + 1] = fx;
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[2 * dy] = 1.f - fy;
        beta[2 * dy + 1] = fy;
    }
    parallel_rows(dh, nthreads, [&](int ya, int yb) {
        std::vector<float> r0((size_t)dw * cn), r1((size_t)dw * cn);
        for (int dy = ya; dy < yb; dy++) {
            int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);
            int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
            const float* S0 = src + (size_t)sy0 * sw * cn;
            const float* S1 = src + (size_t)sy1 * sw * cn;
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx];
                int sx1 = std::min(sx + 1, sw - 1);  // weight is 0 whenever sx+1 is outside
                float a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
                for (int c = 0; c < cn; c++) {
                    r0[dx * cn + c] = S0[sx * cn + c] * a0 + S0[sx1 * cn + c] * a1;
                    r1[dx * cn + c] = S1[sx * cn + c] * a0 + S1[sx1 * cn + c] * a1;
                }
            }
            float b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
            float* D = dst + (size_t)dy * dw * cn;
            for (int i = 0; i < dw * cn; i++) D[i] = r0[i] * b0 + r1[i] * b1;
        }
    });
}

// ---------------------------------------------------------------- optflow.cpp
// FarnebackPrepareGaussian
void prepare_gaussian(int n, double sigma, float* g, float* xg, float* xxg, double& ig11,
                      double& ig03, double& ig33, double& ig55) {
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[6][6] = {{0}};
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            // float products, accumulated in double, exactly as `G(0,0) += g[y]*g[x]`
            G[0][0] += g[y] * g[x];
            G[1][1] += g[y] * g[x] * x * x;
            G[3][3] += g[y] * g[x] * x * x * x * x;
            G[5][5] += g[y] * g[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    // invG = G.inv(DECOMP_CHOLESKY): cv::invert -> setIdentity(dst); hal::Cholesky64f(G, dst)
    // (core/src/matrix_decomp.cpp CholImpl<double>; a 6x6 is below the LAPACK threshold of
    // hal_internal.cpp).  L keeps 1/sqrt(pivot) on its diagonal; forward then backward
    // substitution on the identity, every partial sum in double.
    double L[6][6], B[6][6];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            L[i][j] = G[i][j];
            B[i][j] = i == j ? 1. : 0.;
        }
    for (int i = 0; i < 6; i++) {
        double s;
        int j, k;
        for (j = 0; j < i; j++) {
            s = L[i][j];
            for (k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            L[i][j] = s * L[j][j];
        }
        s = L[i][i];
        for (k = 0; k < j; k++) {
            double t = L[i][k];
            s -= t * t;
        }
        L[i][i] = 1. / std::sqrt(s);
    }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double s = B[i][j];
            for (int k = 0; k < i; k++) s -= L[i][k] * B[k][j];
            B[i][j] = s * L[i][i];
        }
    for (int i = 5; i >= 0; i--)
        for (int j = 0; j < 6; j++) {
            double s = B[i][j];
            for (int k = 5; k > i; k--) s -= L[k][i] * B[k][j];
            B[i][j] = s * L[i][i];
        }
    ig11 = B[1][1];
    ig03 = B[0][3];
    ig33 = B[3][3];
    ig55 = B[5][5];
}

// FarnebackPolyExp: R = (y, x, y^2, x^2, xy) coefficients, 5 interleaved floats/px.
void polyexp(const float* src, int width, int height, int n, double sigma, float* dst,
             int nthreads) {
    std::vector<float> kbuf(n * 6 + 3);
    float* g = kbuf.data() + n;
    float* xg = g + n * 2 + 1;
    float* xxg = xg + n * 2 + 1;
    double ig11, ig03, ig33, ig55;
    prepare_gaussian(n, sigma, g, xg, xxg, ig11, ig03, ig33, ig55);

    parallel_rows(height, nthreads, [&](int ya, int yb) {
        std::vector<float> _row((size_t)(width + n * 2) * 3);
        float* row = _row.data() + n * 3;
        for (int y = ya; y < yb; y++) {
            float g0 = g[0], g1, g2;
            const float* srow0 = src + (size_t)y * width;
            const float* srow1 = nullptr;
            float* drow = dst + (size_t)y * width * 5;

            // vertical part of convolution (float accumulators)
            for (int x = 0; x < width; x++) {
                row[x * 3] = srow0[x] * g0;
                row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
            }
            for (int k = 1; k <= n; k++) {
                g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
                srow0 = src + (size_t)std::max(y - k, 0) * width;
                srow1 = src + (size_t)std::min(y + k, height - 1) * width;
                for (int x = 0; x < width; x++) {
                    float p = srow0[x] + srow1[x];
                    float t0 = row[x * 3] + g0 * p;
                    float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                    float t2 = row[x * 3 + 2] + g2 * p;
                    row[x * 3] = t0;
                    row[x * 3 + 1] = t1;
                    row[x * 3 + 2] = t2;
                }
            }
            // horizontal part of convolution: replicate the first/last triplet
            for (int x = 0; x < n * 3; x++) {
                row[-1 - x] = row[2 - x];
                row[width * 3 + x] = row[(width - 1) * 3 + x];
            }
            for (int x = 0; x < width; x++) {
                g0 = g[0];
                // b1 ~ 1, b2 ~ x, b3 ~ y, b4 ~ x^2, b5 ~ y^2, b6 ~ xy (double accumulators;
                // the float sub-expressions round to float first, as in the C++ source)
                double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0,
                       b5 = row[x * 3 + 2] * g0, b6 = 0;
                for (int k = 1; k <= n; k++) {
                    double tg = row[(x + k) * 3] + row[(x - k) * 3];
                    g0 = g[k];
                    b1 += tg * g0;
                    b4 += tg * xxg[k];
                    b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                    b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                    b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                    b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
                }
                // do not store r1
                drow[x * 5 + 1] = (float)(b2 * ig11);
                drow[x * 5] = (float)(b3 * ig11);
                drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
                drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
                drow[x * 5 + 4] = (float)(b6 * ig55);
            }
        }
    });
}

// FarnebackUpdateMatrices
void update_matrices(const float* R0_, const float* R1, const float* flow_, float* M_,
                     int width, int height, int y0_, int y1_) {
    const int BORDER = 5;
    static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    const size_t step1 = (size_t)width * 5;
    for (int y = y0_; y < y1_; y++) {
        const float* flow = flow_ + (size_t)y * width * 2;
        const float* R0 = R0_ + (size_t)y * width * 5;
        float* M = M_ + (size_t)y * width * 5;
        for (int x = 0; x < width; x++) {
            float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = cv_floor(fx), y1 = cv_floor(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1;
            fy -= y1;
            if ((unsigned)x1 < (unsigned)(width - 1) && (unsigned)y1 < (unsigned)(height - 1)) {
                const float* ptr = R1 + (size_t)y1 * step1 + (size_t)x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy),
                      a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0[x * 5 + 2];
                r5 = R0[x * 5 + 3];
                r6 = R0[x * 5 + 4] * 0.5f;
            }
            r2 = (R0[x * 5] - r2) * 0.5f;
            r3 = (R0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(width - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(height - BORDER * 2)) {
                float scale = (x < BORDER ? border[x] : 1.f) *
                              (x >= width - BORDER ? border[width - x - 1] : 1.f) *
                              (y < BORDER ? border[y] : 1.f) *
                              (y >= height - BORDER ? border[height - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            M[x * 5] = r4 * r4 + r6 * r6;      // G(1,1)
            M[x * 5 + 1] = (r4 + r5) * r6;     // G(1,2)=G(2,1)
            M[x * 5 + 2] = r5 * r5 + r6 * r6;  // G(2,2)
            M[x * 5 + 3] = r4 * r2 + r6 * r3;  // h(1)
            M[x * 5 + 4] = r6 * r2 + r5 * r3;  // h(2)
        }
    }
}

// FarnebackUpdateFlow_Blur (box window, double running sums, literal stripe logic)
// gout (optional, 3 floats per pixel): the window values (g11, g12, g22) the solve used.
void update_flow_blur(const float* R0, const float* R1, float* flow_, float* matM, int width,
                      int height, int block_size, bool update, float* gout = nullptr) {
    int m = block_size / 2;
    int y0 = 0, y1;
    int min_update_stripe = std::max((1 << 10) / width, block_size);
    double scale = 1. / (block_size * block_size);
    std::vector<double> _vsum((size_t)(width + m * 2 + 2) * 5);
    double* vsum = _vsum.data() + (m + 1) * 5;

    const float* srow0 = matM;
    for (int x = 0; x < width * 5; x++) vsum[x] = srow0[x] * (m + 2);
    for (int y = 1; y < m; y++) {
        srow0 = matM + (size_t)std::min(y, height - 1) * width * 5;
        for (int x = 0; x < width * 5; x++) vsum[x] += srow0[x];
    }
    for (int y = 0; y < height; y++) {
        double g11, g12, g22, h1, h2;
        float* flow = flow_ + (size_t)y * width * 2;
        srow0 = matM + (size_t)std::max(y - m - 1, 0) * width * 5;
        const float* srow1 = matM + (size_t)std::min(y + m, height - 1) * width * 5;
        // vertical blur: float subtraction, double accumulation
        for (int x = 0; x < width * 5; x++) vsum[x] += srow1[x] - srow0[x];
        // update borders
        for (int x = 0; x < (m + 1) * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[width * 5 + x] = vsum[width * 5 + x - 5];
        }
        g11 = vsum[0] * (m + 2);
        g12 = vsum[1] * (m + 2);
        g22 = vsum[2] * (m + 2);
        h1 = vsum[3] * (m + 2);
        h2 = vsum[4] * (m + 2);
        for (int x = 1; x < m; x++) {
            g11 += vsum[x * 5];
            g12 += vsum[x * 5 + 1];
            g22 += vsum[x * 5 + 2];
            h1 += vsum[x * 5 + 3];
            h2 += vsum[x * 5 + 4];
        }
        for (int x = 0; x < width; x++) {
            g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
            g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
            g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
            h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
            h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
            double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale;
            double h1_ = h1 * scale, h2_ = h2 * scale;
            double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
            flow[x * 2] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
            flow[x * 2 + 1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
            if (gout) {
                float* go = gout + ((size_t)y * width + x) * 3;
                go[0] = (float)g11_; go[1] = (float)g12_; go[2] = (float)g22_;
            }
        }
        y1 = y == height - 1 ? height : y - block_size;
        if (update && (y1 == height || y1 >= y0 + min_update_stripe)) {
            update_matrices(R0, R1, flow_, matM, width, height, y0, y1);
            y0 = y1;
        }
    }
}

// FarnebackUpdateFlow_GaussianBlur (float separable window, literal stripe logic)
void update_flow_gaussian(const float* R0, const float* R1, float* flow_, float* matM,
                          int width, int height, int block_size, bool update, float* gout = nullptr) {
    int m = block_size / 2;
    int y0 = 0, y1;
    int min_update_stripe = std::max((1 << 10) / width, block_size);
    double sigma = m * 0.3, s = 1;
    std::vector<float> _vsum((size_t)(width + m * 2 + 2) * 5), hsum((size_t)width * 5);
    std::vector<float> kernel(m + 1);
    std::vector<const float*> srow(m * 2 + 1);
    float* vsum = _vsum.data() + (m + 1) * 5;
    kernel[0] = (float)s;
    for (int i = 1; i <= m; i++) {
        float t = (float)std::exp(-i * i / (2 * sigma * sigma));
        kernel[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; i++) kernel[i] = (float)(kernel[i] * s);

    for (int y = 0; y < height; y++) {
        double g11, g12, g22, h1, h2;
        float* flow = flow_ + (size_t)y * width * 2;
        for (int i = 0; i <= m; i++) {
            srow[m - i] = matM + (size_t)std::max(y - i, 0) * width * 5;
            srow[m + i] = matM + (size_t)std::min(y + i, height - 1) * width * 5;
        }
        for (int x = 0; x < width * 5; x++) {
            float s0 = srow[m][x] * kernel[0];
            for (int i = 1; i <= m; i++) s0 += (srow[m + i][x] + srow[m - i][x]) * kernel[i];
            vsum[x] = s0;
        }
        for (int x = 0; x < m * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[width * 5 + x] = vsum[width * 5 + x - 5];
        }
        for (int x = 0; x < width * 5; x++) {
            float sum = vsum[x] * kernel[0];
            for (int i = 1; i <= m; i++) sum += kernel[i] * (vsum[x - i * 5] + vsum[x + i * 5]);
            hsum[x] = sum;
        }
        for (int x = 0; x < width; x++) {
            g11 = hsum[x * 5];
            g12 = hsum[x * 5 + 1];
            g22 = hsum[x * 5 + 2];
            h1 = hsum[x * 5 + 3];
            h2 = hsum[x * 5 + 4];
            double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            flow[x * 2] = (float)((g11 * h2 - g12 * h1) * idet);
            flow[x * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
            if (gout) {
                float* go = gout + ((size_t)y * width + x) * 3;
                go[0] = (float)g11; go[1] = (float)g12; go[2] = (float)g22;
            }
        }
        y1 = y == height - 1 ? height : y - block_size;
        if (update && (y1 == height || y1 >= y0 + min_update_stripe)) {
            update_matrices(R0, R1, flow_, matM, width, height, y0, y1);
            y0 = y1;
        }
    }
}

// Multi-threaded forms for the CPU baseline on all cores and for the full-size GPU tests (a 4K pair in a second
// instead of three).  They rely on the equivalence the CPU tier proves (test_stripe_update_equals_whole_image):
// upstream's in-loop stripe update of M equals "window + solve the whole image, then update every matrix".
// Box window: every column's running sum is its own sequential chain (threads split the columns), every row's
// running sum likewise (threads split the rows), so each chain sees exactly the operations of the sequential
// code above: same bits (oracle_kat compares 1 and 3 threads).
void update_flow_blur_mt(const float* R0, const float* R1, float* flow_, float* matM, int width, int height,
                         int block_size, bool update, int nthreads, float* gout, std::vector<double>& V) {
    const int m = block_size / 2;
    const double scale = 1. / (block_size * block_size);
    const size_t row5 = (size_t)width * 5;
    if (V.size() < (size_t)height * row5) V.resize((size_t)height * row5);     // the caller keeps it across scales and iterations
    parallel_rows(width, nthreads, [&](int xa, int xb) {      // column ranges
        const size_t a5 = (size_t)xa * 5, b5 = (size_t)xb * 5;
        std::vector<double> vs(b5 - a5);
        for (size_t x = a5; x < b5; x++) vs[x - a5] = matM[x] * (m + 2);
        for (int y = 1; y < m; y++) {
            const float* r = matM + (size_t)std::min(y, height - 1) * row5;
            for (size_t x = a5; x < b5; x++) vs[x - a5] += r[x];
        }
        for (int y = 0; y < height; y++) {
            const float* s0 = matM + (size_t)std::max(y - m - 1, 0) * row5;
            const float* s1 = matM + (size_t)std::min(y + m, height - 1) * row5;
            double* o = V.data() + (size_t)y * row5;
            for (size_t x = a5; x < b5; x++) {
                vs[x - a5] += s1[x] - s0[x];
                o[x] = vs[x - a5];
            }
        }
    });
    parallel_rows(height, nthreads, [&](int ya, int yb) {
        std::vector<double> _vsum((size_t)(width + m * 2 + 2) * 5);
        double* vsum = _vsum.data() + (m + 1) * 5;
        for (int y = ya; y < yb; y++) {
            std::memcpy(vsum, V.data() + (size_t)y * row5, sizeof(double) * row5);
            for (int x = 0; x < (m + 1) * 5; x++) {
                vsum[-1 - x] = vsum[4 - x];
                vsum[width * 5 + x] = vsum[width * 5 + x - 5];
            }
            double g11 = vsum[0] * (m + 2), g12 = vsum[1] * (m + 2), g22 = vsum[2] * (m + 2);
            double h1 = vsum[3] * (m + 2), h2 = vsum[4] * (m + 2);
            for (int x = 1; x < m; x++) {
                g11 += vsum[x * 5]; g12 += vsum[x * 5 + 1]; g22 += vsum[x * 5 + 2];
                h1 += vsum[x * 5 + 3]; h2 += vsum[x * 5 + 4];
            }
            float* flow = flow_ + (size_t)y * width * 2;
            for (int x = 0; x < width; x++) {
                g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
                g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
                g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
                h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
                h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
                double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale;
                double h1_ = h1 * scale, h2_ = h2 * scale;
                double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
                flow[x * 2] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
                flow[x * 2 + 1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
                if (gout) {
                    float* go = gout + ((size_t)y * width + x) * 3;
                    go[0] = (float)g11_; go[1] = (float)g12_; go[2] = (float)g22_;
                }
            }
        }
    });
    if (update)
        parallel_rows(height, nthreads, [&](int ya, int yb) { update_matrices(R0, R1, flow_, matM, width, height, ya, yb); });
}

// Gaussian window: rows are independent once M is read-only for the whole pass.
void update_flow_gaussian_mt(const float* R0, const float* R1, float* flow_, float* matM, int width, int height,
                             int block_size, bool update, int nthreads, float* gout) {
    const int m = block_size / 2;
    const double sigma = m * 0.3;
    double s = 1;
    std::vector<float> kernel(m + 1);
    kernel[0] = (float)s;
    for (int i = 1; i <= m; i++) {
        float t = (float)std::exp(-i * i / (2 * sigma * sigma));
        kernel[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; i++) kernel[i] = (float)(kernel[i] * s);
    parallel_rows(height, nthreads, [&](int ya, int yb) {
        std::vector<float> _vsum((size_t)(width + m * 2 + 2) * 5), hsum((size_t)width * 5);
        std::vector<const float*> srow(m * 2 + 1);
        float* vsum = _vsum.data() + (m + 1) * 5;
        for (int y = ya; y < yb; y++) {
            for (int i = 0; i <= m; i++) {
                srow[m - i] = matM + (size_t)std::max(y - i, 0) * width * 5;
                srow[m + i] = matM + (size_t)std::min(y + i, height - 1) * width * 5;
            }
            for (int x = 0; x < width * 5; x++) {
                float s0 = srow[m][x] * kernel[0];
                for (int i = 1; i <= m; i++) s0 += (srow[m + i][x] + srow[m - i][x]) * kernel[i];
                vsum[x] = s0;
            }
            for (int x = 0; x < m * 5; x++) {
                vsum[-1 - x] = vsum[4 - x];
                vsum[width * 5 + x] = vsum[width * 5 + x - 5];
            }
            for (int x = 0; x < width * 5; x++) {
                float sum = vsum[x] * kernel[0];
                for (int i = 1; i <= m; i++) sum += kernel[i] * (vsum[x - i * 5] + vsum[x + i * 5]);
                hsum[x] = sum;
            }
            float* flow = flow_ + (size_t)y * width * 2;
            for (int x = 0; x < width; x++) {
                double g11 = hsum[x * 5], g12 = hsum[x * 5 + 1], g22 = hsum[x * 5 + 2], h1 = hsum[x * 5 + 3], h2 = hsum[x * 5 + 4];
                double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
                flow[x * 2] = (float)((g11 * h2 - g12 * h1) * idet);
                flow[x * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
                if (gout) {
                    float* go = gout + ((size_t)y * width + x) * 3;
                    go[0] = (float)g11; go[1] = (float)g12; go[2] = (float)g22;
                }
            }
        }
    });
    if (update)
        parallel_rows(height, nthreads, [&](int ya, int yb) { update_matrices(R0, R1, flow_, matM, width, height, ya, yb); });
}

struct LevelGeom {
    int w, h, ksize;
    double sigma, scale;
};

int crop_levels(int w, int h, double pyr_scale, int levels) {
    const int min_size = 32;
    int k;
    double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (w * scale < min_size || h * scale < min_size) break;
    }
    return k;
}

LevelGeom level_geom(int w, int h, double pyr_scale, int k) {
    LevelGeom g;
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= pyr_scale;
    g.scale = scale;
    g.sigma = (1. / scale - 1) * 0.5;
    int smooth_sz = cv_round(g.sigma * 5) | 1;
    g.ksize = std::max(smooth_sz, 3);
    g.w = cv_round(w * scale);
    g.h = cv_round(h * scale);
    return g;
}

void pyr_level(const uint8_t* img, size_t step, int w, int h, double sigma, int ksize,
               float* out, int ow, int oh, int nthreads) {
    std::vector<float> fimg((size_t)w * h), blurred((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) fimg[(size_t)y * w + x] = (float)img[y * step + x];
    gaussian_blur_f32(fimg.data(), w, h, ksize, sigma, blurred.data(), nthreads);
    resize_linear_f32(blurred.data(), w, h, 1, out, ow, oh, nthreads);
}

}  // namespace

// ------------------------------------------------------------------ C interface
extern "C" {

int orc_level_geometry(int w, int h, double pyr_scale, int levels, int k, int* wk, int* hk,
                       double* sigma, int* ksize) {
    int L = crop_levels(w, h, pyr_scale, levels);
    LevelGeom g = level_geom(w, h, pyr_scale, k);
    if (wk) *wk = g.w;
    if (hk) *hk = g.h;
    if (sigma) *sigma = g.sigma;
    if (ksize) *ksize = g.ksize;
    return L;
}

int orc_gaussian_kernel(int n, double sigma, float* k) {
    gaussian_kernel(n, sigma, k);
    return 0;
}

int orc_gaussian_blur_f32(const float* src, int w, int h, int ksize, double sigma, float* dst) {
    gaussian_blur_f32(src, w, h, ksize, sigma, dst, 1);
    return 0;
}

int orc_resize_linear_f32(const float* src, int sw, int sh, int cn, float* dst, int dw, int dh) {
    resize_linear_f32(src, sw, sh, cn, dst, dw, dh, 1);
    return 0;
}

int orc_pyr_level(const uint8_t* img, size_t step, int w, int h, double sigma, int ksize,
                  float* out, int ow, int oh) {
    pyr_level(img, step, w, h, sigma, ksize, out, ow, oh, 1);
    return 0;
}

int orc_prepare_gaussian(int n, double sigma, float* g, float* xg, float* xxg, double* ig) {
    prepare_gaussian(n, sigma, g + n, xg + n, xxg + n, ig[0], ig[1], ig[2], ig[3]);
    return 0;
}

int orc_polyexp(const float* I, int w, int h, int n, double sigma, float* R) {
    polyexp(I, w, h, n, sigma, R, 1);
    return 0;
}

int orc_update_matrices(const float* R0, const float* R1, const float* flow, float* M, int w,
                        int h, int y0, int y1) {
    update_matrices(R0, R1, flow, M, w, h, y0, y1);
    return 0;
}

int orc_update_flow_blur(const float* R0, const float* R1, float* flow, float* M, int w, int h,
                         int block_size, int update) {
    update_flow_blur(R0, R1, flow, M, w, h, block_size, update != 0);
    return 0;
}

int orc_update_flow_gaussian(const float* R0, const float* R1, float* flow, float* M, int w,
                             int h, int block_size, int update) {
    update_flow_gaussian(R0, R1, flow, M, w, h, block_size, update != 0);
    return 0;
}

// FarnebackOpticalFlowImpl::calc (CPU path; OPTFLOW_USE_INITIAL_FLOW is never used by
// the reference and is rejected here)
int orc_farneback_u8_ex(const uint8_t* prev, size_t prev_step, const uint8_t* next,
                        size_t next_step, int w, int h, float* flow0, size_t flow_step,
                        double pyr_scale, int levels, int winsize, int iters, int poly_n,
                        double poly_sigma, int flags, int nthreads, const orc_farneback_diag* dg) {
    if (!prev || !next || !flow0 || w <= 0 || h <= 0 || !(pyr_scale < 1) || pyr_scale <= 0 ||
        poly_n < 1 || winsize < 1 || iters < 0 || (flags & ~ORC_FARNEBACK_GAUSSIAN))
        return -1;
    const uint8_t* img[2] = {prev, next};
    const size_t steps[2] = {prev_step, next_step};
    levels = crop_levels(w, h, pyr_scale, levels);

    std::vector<float> prevFlow, flow;
    std::vector<double> Vscratch;
    if (nthreads > 1 && !(flags & ORC_FARNEBACK_GAUSSIAN)) Vscratch.resize((size_t)w * h * 5);    // scale 0 is the largest
    int pw = 0, ph = 0;
    for (int k = levels; k >= 0; k--) {
        LevelGeom g = level_geom(w, h, pyr_scale, k);
        int width = g.w, height = g.h;
        flow.assign((size_t)width * height * 2, 0.f);
        if (!prevFlow.empty()) {
            resize_linear_f32(prevFlow.data(), pw, ph, 2, flow.data(), width, height, nthreads);
            float a = (float)(1. / pyr_scale);  // Mat *= double on CV_32F: float multiply
            for (auto& v : flow) v = v * a;
        }
        std::vector<float> R[2], I((size_t)width * height), M((size_t)width * height * 5);
        for (int i = 0; i < 2; i++) {
            pyr_level(img[i], steps[i], w, h, g.sigma, g.ksize, I.data(), width, height, nthreads);
            R[i].resize((size_t)width * height * 5);
            polyexp(I.data(), width, height, poly_n, poly_sigma, R[i].data(), nthreads);
        }
        parallel_rows(height, nthreads, [&](int ya, int yb) {
            update_matrices(R[0].data(), R[1].data(), flow.data(), M.data(), width, height, ya, yb);
        });
        const bool want_g = dg && (dg->det_min || (k == 0 && dg->g_last));
        std::vector<float> gbuf(want_g ? (size_t)width * height * 3 : 0);
        std::vector<float> dmin(dg && dg->det_min ? (size_t)width * height : 0, FLT_MAX);
        for (int i = 0; i < iters; i++) {
            float* gp = want_g ? gbuf.data() : nullptr;
            const bool mt = nthreads > 1 && height >= 2 * nthreads && width >= 2 * nthreads;
            if (flags & ORC_FARNEBACK_GAUSSIAN) {
                if (mt) update_flow_gaussian_mt(R[0].data(), R[1].data(), flow.data(), M.data(), width, height, winsize, i < iters - 1, nthreads, gp);
                else update_flow_gaussian(R[0].data(), R[1].data(), flow.data(), M.data(), width, height, winsize, i < iters - 1, gp);
            } else {
                if (mt) update_flow_blur_mt(R[0].data(), R[1].data(), flow.data(), M.data(), width, height, winsize, i < iters - 1, nthreads, gp, Vscratch);
                else update_flow_blur(R[0].data(), R[1].data(), flow.data(), M.data(), width, height, winsize, i < iters - 1, gp);
            }
            for (size_t p = 0; p < dmin.size(); p++) {
                const float* go = gbuf.data() + p * 3;
                dmin[p] = std::min(dmin[p], (float)((double)go[0] * go[2] - (double)go[1] * go[1]));
            }
        }
        if (dg && dg->g_last && k == 0 && iters > 0)
            std::memcpy(dg->g_last, gbuf.data(), sizeof(float) * gbuf.size());
        if (dg && dg->det_min && iters > 0) {
            // a full-resolution pixel inherits the worst determinant of the 3x3 block around its
            // ancestor at this scale (the footprint of the bilinear up-sampling and of a small window)
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    int ax = std::min((int)((x + 0.5) * g.scale), width - 1);
                    int ay = std::min((int)((y + 0.5) * g.scale), height - 1);
                    float v = FLT_MAX;
                    for (int yy = std::max(ay - 1, 0); yy <= std::min(ay + 1, height - 1); yy++)
                        for (int xx = std::max(ax - 1, 0); xx <= std::min(ax + 1, width - 1); xx++)
                            v = std::min(v, dmin[(size_t)yy * width + xx]);
                    float& o = dg->det_min[(size_t)y * w + x];
                    o = (k == levels) ? v : std::min(o, v);
                }
        }
        if (dg && k < ORC_MAX_DIAG_LEVELS && dg->level_flow[k])
            std::memcpy(dg->level_flow[k], flow.data(), sizeof(float) * flow.size());
        prevFlow.swap(flow);
        pw = width;
        ph = height;
    }
    for (int y = 0; y < h; y++)
        std::memcpy((char*)flow0 + y * flow_step, prevFlow.data() + (size_t)y * w * 2,
                    sizeof(float) * 2 * w);
    return 0;
}

int orc_farneback_u8(const uint8_t* prev, size_t prev_step, const uint8_t* next,
                     size_t next_step, int w, int h, float* flow0, size_t flow_step,
                     double pyr_scale, int levels, int winsize, int iters, int poly_n,
                     double poly_sigma, int flags, int nthreads) {
    return orc_farneback_u8_ex(prev, prev_step, next, next_step, w, h, flow0, flow_step, pyr_scale,
                               levels, winsize, iters, poly_n, poly_sigma, flags, nthreads, nullptr);
}

}  // extern "C"
