// lk_kernels.hip -- sparse pyramidal Lucas-Kanade for gfx950.
//
// Replaces cv::calcOpticalFlowPyrLK at Streakline.cpp:32, ripcurrents_module.cpp:716, :738, :775,
// :1162 (8UC1 images, <= a few hundred points).  Arithmetic follows OpenCV 4.1.0 lkpyramid.cpp:
// 8-bit pyramid by pyrDown (REFLECT_101), Scharr derivatives in int16, W_BITS = 14 fixed-point
// bilinear weights, 2x2 system per point, <= maxCount Newton steps per level.  The window sums
// are exact integers here (int64 partial sums, tree-reduced) where upstream adds floats in raster
// order; everything else is operation for operation.
//
//   k_lk_pyrdown   imgproc pyrDown 8U: (1 4 6 4 1)^2, (sum + 128) >> 8         one thread / pixel
//   k_lk_scharr    calcSharrDeriv: (dx, dy) int16 interleaved                  one thread / pixel
//   k_lk_track     LKTrackerInvoker, all levels of one point                   one block / point
//
// No padded copies: image reads outside a level reflect (== the REFLECT_101 border
// buildOpticalFlowPyramid adds), derivative reads outside are 0 (== its BORDER_CONSTANT).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "../../include/rcflow.h"
#include "rc_device.h"
#include "rc_host.h"

#define RC_LK_MAX_LEVELS 8
#define RC_LK_THREADS 256

struct RcLkLevel {
    const uint8_t* I;      // prev level, pitch = w
    const int16_t* dI;     // its Scharr derivatives, (dx, dy) interleaved
    const uint8_t* J;      // next level
    int w, h;
};
struct RcLkArgs {
    RcLkLevel lv[RC_LK_MAX_LEVELS];
    int max_level;
    const float2* prev_pts;
    float2* next_pts;
    uint8_t* status;
    float* err;
    int npts, win_w, win_h, max_count, flags;
    double epsilon;        // already squared
    double min_eig;
};

__global__ __launch_bounds__(RC_BLOCK) void k_lk_pyrdown(const uint8_t* __restrict__ src, size_t step, int sw, int sh,
                                                         uint8_t* __restrict__ dst, int dw, int dh) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    int cx[5];
#pragma unroll
    for (int j = 0; j < 5; j++) cx[j] = rc_reflect101(2 * x - 2 + j, sw);
    int r[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint8_t* s = src + (size_t)rc_reflect101(2 * y - 2 + k, sh) * step;
        r[k] = s[cx[2]] * 6 + (s[cx[1]] + s[cx[3]]) * 4 + s[cx[0]] + s[cx[4]];
    }
    dst[(size_t)y * dw + x] = (uint8_t)((r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4] + 128) >> 8);
}

__global__ __launch_bounds__(RC_BLOCK) void k_lk_scharr(const uint8_t* __restrict__ src, size_t step, int w, int h,
                                                        int16_t* __restrict__ dxy) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint8_t* s0 = src + (size_t)rc_reflect101(y - 1, h) * step;
    const uint8_t* s1 = src + (size_t)y * step;
    const uint8_t* s2 = src + (size_t)rc_reflect101(y + 1, h) * step;
    const int xl = rc_reflect101(x - 1, w), xr = rc_reflect101(x + 1, w);
    // trow0 = (up + down) * 3 + mid * 10, trow1 = down - up at columns x-1, x, x+1
    const int a_l = (s0[xl] + s2[xl]) * 3 + s1[xl] * 10, a_r = (s0[xr] + s2[xr]) * 3 + s1[xr] * 10;
    const int b_l = s2[xl] - s0[xl], b_c = s2[x] - s0[x], b_r = s2[xr] - s0[xr];
    short2 d;
    d.x = (short)(a_r - a_l);
    d.y = (short)((b_r + b_l) * 3 + b_c * 10);
    *(short2*)(dxy + ((size_t)y * w + x) * 2) = d;
}

__device__ __forceinline__ int rc_lk_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int rc_lk_round(float v) { return (int)rintf(v); }      // cvRound: half to even

struct RcLkWeights { int w00, w01, w10, w11; };
__device__ __forceinline__ RcLkWeights rc_lk_weights(float a, float b) {
    RcLkWeights q;
    q.w00 = rc_lk_round((1.f - a) * (1.f - b) * (float)(1 << 14));
    q.w01 = rc_lk_round(a * (1.f - b) * (float)(1 << 14));
    q.w10 = rc_lk_round((1.f - a) * b * (float)(1 << 14));
    q.w11 = (1 << 14) - q.w00 - q.w01 - q.w10;
    return q;
}
// bilinear sample of the 8-bit level at integer corner (x, y), scaled by 2^5 (W_BITS1 - 5 = 9)
__device__ __forceinline__ int rc_lk_sample_u8(const uint8_t* __restrict__ img, int w, int h, int x, int y,
                                               const RcLkWeights& q) {
    const int x0 = rc_reflect101(x, w), x1 = rc_reflect101(x + 1, w);
    const uint8_t* r0 = img + (size_t)rc_reflect101(y, h) * w;
    const uint8_t* r1 = img + (size_t)rc_reflect101(y + 1, h) * w;
    return rc_lk_descale(r0[x0] * q.w00 + r0[x1] * q.w01 + r1[x0] * q.w10 + r1[x1] * q.w11, 9);
}
__device__ __forceinline__ short2 rc_lk_deriv_at(const int16_t* __restrict__ d, int w, int h, int x, int y) {
    if ((unsigned)x >= (unsigned)w || (unsigned)y >= (unsigned)h) return make_short2(0, 0);
    return *(const short2*)(d + ((size_t)y * w + x) * 2);
}

// sums three int64 per thread over the block; every thread returns with the totals
__device__ __forceinline__ void rc_lk_reduce3(long long* red, long long& a, long long& b, long long& c) {
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = a; red[RC_LK_THREADS + tid] = b; red[2 * RC_LK_THREADS + tid] = c;
    __syncthreads();
    for (int s = RC_LK_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] += red[tid + s];
            red[RC_LK_THREADS + tid] += red[RC_LK_THREADS + tid + s];
            red[2 * RC_LK_THREADS + tid] += red[2 * RC_LK_THREADS + tid + s];
        }
        __syncthreads();
    }
    a = red[0]; b = red[RC_LK_THREADS]; c = red[2 * RC_LK_THREADS];
}

__global__ __launch_bounds__(RC_LK_THREADS) void k_lk_track(RcLkArgs a) {
    extern __shared__ __align__(16) unsigned char lk_smem[];
    const int win_w = a.win_w, win_h = a.win_h, win_n = win_w * win_h;
    long long* red = (long long*)lk_smem;                                  // [3][RC_LK_THREADS]
    short* Ibuf = (short*)(lk_smem + 3 * RC_LK_THREADS * sizeof(long long)); // [win_n]
    short2* dIbuf = (short2*)(Ibuf + ((win_n + 1) & ~1));                   // [win_n]
    const int pt = blockIdx.x, tid = threadIdx.x;
    if (pt >= a.npts) return;
    const bool use_initial = (a.flags & 4) != 0, get_min_eig = (a.flags & 8) != 0;
    const float halfx = (win_w - 1) * 0.5f, halfy = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float2 p0 = a.prev_pts[pt];
    float2 nxt = a.next_pts[pt];       // read by every thread before anyone writes (see the barrier in reduce3)
    bool ok = true;
    float errv = 0.f;

    for (int level = a.max_level; level >= 0; level--) {
        const RcLkLevel L = a.lv[level];
        const float sc = (float)(1. / (double)(1 << level));
        float px = p0.x * sc, py = p0.y * sc;
        float nx, ny;
        if (level == a.max_level) {
            if (use_initial) { nx = nxt.x * sc; ny = nxt.y * sc; }
            else { nx = px; ny = py; }
        } else {
            nx = nxt.x * 2.f; ny = nxt.y * 2.f;
        }
        nxt = make_float2(nx, ny);
        px -= halfx; py -= halfy;
        const int ipx = rc_cvt_i32_x86(floorf(px)), ipy = rc_cvt_i32_x86(floorf(py));
        if (ipx < -win_w || ipx >= L.w || ipy < -win_h || ipy >= L.h) {
            if (level == 0) { ok = false; errv = 0.f; }
            continue;
        }
        // patch of the first image + covariance of its derivatives
        const RcLkWeights q = rc_lk_weights(px - ipx, py - ipy);
        long long sA11 = 0, sA12 = 0, sA22 = 0;
        for (int idx = tid; idx < win_n; idx += RC_LK_THREADS) {
            const int y = idx / win_w, x = idx - y * win_w;
            const int gx = ipx + x, gy = ipy + y;
            const int ival = rc_lk_sample_u8(L.I, L.w, L.h, gx, gy, q);
            const short2 d00 = rc_lk_deriv_at(L.dI, L.w, L.h, gx, gy), d01 = rc_lk_deriv_at(L.dI, L.w, L.h, gx + 1, gy);
            const short2 d10 = rc_lk_deriv_at(L.dI, L.w, L.h, gx, gy + 1), d11 = rc_lk_deriv_at(L.dI, L.w, L.h, gx + 1, gy + 1);
            const int ixval = rc_lk_descale(d00.x * q.w00 + d01.x * q.w01 + d10.x * q.w10 + d11.x * q.w11, 14);
            const int iyval = rc_lk_descale(d00.y * q.w00 + d01.y * q.w01 + d10.y * q.w10 + d11.y * q.w11, 14);
            Ibuf[idx] = (short)ival;
            dIbuf[idx] = make_short2((short)ixval, (short)iyval);
            sA11 += (long long)ixval * ixval;
            sA12 += (long long)ixval * iyval;
            sA22 += (long long)iyval * iyval;
        }
        rc_lk_reduce3(red, sA11, sA12, sA22);
        const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win_w * win_h);
        if (get_min_eig) errv = minEig;
        if ((double)minEig < a.min_eig || D < 1.1920928955078125e-07f) {
            if (level == 0) ok = false;
            continue;
        }
        D = 1.f / D;
        nx -= halfx; ny -= halfy;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; j++) {
            const int inx = rc_cvt_i32_x86(floorf(nx)), iny = rc_cvt_i32_x86(floorf(ny));
            if (inx < -win_w || inx >= L.w || iny < -win_h || iny >= L.h) {
                if (level == 0) ok = false;
                break;
            }
            const RcLkWeights r = rc_lk_weights(nx - inx, ny - iny);
            long long sb1 = 0, sb2 = 0, unused = 0;
            for (int idx = tid; idx < win_n; idx += RC_LK_THREADS) {
                const int y = idx / win_w, x = idx - y * win_w;
                const int diff = rc_lk_sample_u8(L.J, L.w, L.h, inx + x, iny + y, r) - Ibuf[idx];
                const short2 d = dIbuf[idx];
                sb1 += (long long)diff * d.x;
                sb2 += (long long)diff * d.y;
            }
            rc_lk_reduce3(red, sb1, sb2, unused);
            const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            nxt = make_float2(nx + halfx, ny + halfy);
            if ((double)dx * dx + (double)dy * dy <= a.epsilon) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01 && fabsf(dy + pdy) < 0.01) {
                nxt.x -= dx * 0.5f;
                nxt.y -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (ok && level == 0 && !get_min_eig) {
            // L1 residual of the final position (the err output without GET_MIN_EIGENVALS)
            const float fx = nxt.x - halfx, fy = nxt.y - halfy;
            const int inx = rc_cvt_i32_x86(floorf(fx)), iny = rc_cvt_i32_x86(floorf(fy));
            if (inx < -win_w || inx >= L.w || iny < -win_h || iny >= L.h) {
                ok = false;
            } else {
                const RcLkWeights r = rc_lk_weights(fx - inx, fy - iny);
                long long se = 0, u1 = 0, u2 = 0;
                for (int idx = tid; idx < win_n; idx += RC_LK_THREADS) {
                    const int y = idx / win_w, x = idx - y * win_w;
                    const int diff = rc_lk_sample_u8(L.J, L.w, L.h, inx + x, iny + y, r) - Ibuf[idx];
                    se += diff < 0 ? -diff : diff;
                }
                rc_lk_reduce3(red, se, u1, u2);
                errv = (float)se * 1.f / (float)(32 * win_w * win_h);
            }
        }
    }
    if (tid == 0) {
        a.next_pts[pt] = nxt;
        a.status[pt] = ok ? 1 : 0;
        if (a.err) a.err[pt] = errv;
    }
}

static int lk_levels(int w, int h, int win_w, int win_h, int max_level) {
    int cw = w, ch = h;
    for (int level = 0; level <= max_level; level++) {
        int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
        if (nw <= win_w || nh <= win_h) return level;
        cw = nw; ch = nh;
    }
    return max_level;
}

extern "C" int rcflow_pyrlk_levels(int w, int h, int win_w, int win_h, int max_level) {
    if (w < 1 || h < 1 || win_w < 3 || win_h < 3 || max_level < 0) return RC_EINVAL;
    return lk_levels(w, h, win_w, win_h, max_level);
}

extern "C" int rcflow_pyrlk_dev(rc_ctx* ctx, int stream, const uint8_t* d_prev, size_t prev_step, const uint8_t* d_next,
                                size_t next_step, int w, int h, const float* d_prev_pts, float* d_next_pts, int npts,
                                uint8_t* d_status, float* d_err, int win_w, int win_h, int max_level, int crit_type,
                                int max_count, double epsilon, int flags, double min_eig_threshold) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!d_prev || !d_next || npts < 0 || (npts > 0 && (!d_prev_pts || !d_next_pts || !d_status)) || w < 1 || h < 1 ||
        prev_step < (size_t)w || next_step < (size_t)w || win_w <= 2 || win_h <= 2 || max_level < 0) {
        rc_set_error("bad PyrLK arguments");
        return RC_EINVAL;
    }
    if (max_level >= RC_LK_MAX_LEVELS || (size_t)win_w * win_h > 128 * 128) {
        rc_set_error("PyrLK: maxLevel < %d and windows up to 128x128 are supported", RC_LK_MAX_LEVELS);
        return RC_EINVAL;
    }
    if (npts == 0) return RC_OK;
    // SparsePyrLKOpticalFlowImpl::calc: criteria clamping, epsilon squared
    if ((crit_type & 1) == 0) max_count = 30;
    else max_count = max_count < 0 ? 0 : (max_count > 100 ? 100 : max_count);
    if ((crit_type & 2) == 0) epsilon = 0.01;
    else epsilon = epsilon < 0. ? 0. : (epsilon > 10. ? 10. : epsilon);
    epsilon *= epsilon;
    RC_HIP(hipSetDevice(ctx->device));
    const int top = lk_levels(w, h, win_w, win_h, max_level);

    // scratch: levels 1..top of both pyramids (u8) and the derivatives of every prev level
    size_t off = 0, offI[RC_LK_MAX_LEVELS] = {0}, offJ[RC_LK_MAX_LEVELS] = {0}, offD[RC_LK_MAX_LEVELS] = {0};
    int lw[RC_LK_MAX_LEVELS], lh[RC_LK_MAX_LEVELS];
    lw[0] = w; lh[0] = h;
    for (int l = 1; l <= top; l++) { lw[l] = (lw[l - 1] + 1) / 2; lh[l] = (lh[l - 1] + 1) / 2; }
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    // level 0 is copied tightly too (pitch = w) so that the tracker has one addressing form
    for (int l = 0; l <= top; l++) {
        offI[l] = take((size_t)lw[l] * lh[l]);
        offJ[l] = take((size_t)lw[l] * lh[l]);
        offD[l] = take((size_t)lw[l] * lh[l] * 4);
    }
    int rc = rc_buf_ensure(s->lk, off);
    if (rc) return rc;
    unsigned char* base = (unsigned char*)s->lk.p;
    RC_HIP(hipMemcpy2DAsync(base + offI[0], w, d_prev, prev_step, w, h, hipMemcpyDeviceToDevice, s->cur));
    RC_HIP(hipMemcpy2DAsync(base + offJ[0], w, d_next, next_step, w, h, hipMemcpyDeviceToDevice, s->cur));
    RcLkArgs a;
    memset(&a, 0, sizeof(a));
    for (int l = 0; l <= top; l++) {
        if (l > 0) {
            dim3 g((lw[l] + 63) / 64, (lh[l] + 3) / 4);
            hipLaunchKernelGGL(k_lk_pyrdown, g, dim3(RC_BLOCK), 0, s->cur, base + offI[l - 1], (size_t)lw[l - 1], lw[l - 1],
                               lh[l - 1], base + offI[l], lw[l], lh[l]);
            hipLaunchKernelGGL(k_lk_pyrdown, g, dim3(RC_BLOCK), 0, s->cur, base + offJ[l - 1], (size_t)lw[l - 1], lw[l - 1],
                               lh[l - 1], base + offJ[l], lw[l], lh[l]);
        }
        dim3 g((lw[l] + 63) / 64, (lh[l] + 3) / 4);
        hipLaunchKernelGGL(k_lk_scharr, g, dim3(RC_BLOCK), 0, s->cur, base + offI[l], (size_t)lw[l], lw[l], lh[l],
                           (int16_t*)(base + offD[l]));
        a.lv[l].I = base + offI[l];
        a.lv[l].J = base + offJ[l];
        a.lv[l].dI = (const int16_t*)(base + offD[l]);
        a.lv[l].w = lw[l]; a.lv[l].h = lh[l];
    }
    a.max_level = top;
    a.prev_pts = (const float2*)d_prev_pts; a.next_pts = (float2*)d_next_pts;
    a.status = d_status; a.err = d_err;
    a.npts = npts; a.win_w = win_w; a.win_h = win_h; a.max_count = max_count; a.flags = flags;
    a.epsilon = epsilon; a.min_eig = min_eig_threshold;
    const int win_n = win_w * win_h;
    size_t lds = 3 * RC_LK_THREADS * sizeof(long long) + (size_t)((win_n + 1) & ~1) * sizeof(short) + (size_t)win_n * sizeof(short2);
    RC_ALLOW_LDS((k_lk_track), lds);
    hipLaunchKernelGGL(k_lk_track, dim3(npts), dim3(RC_LK_THREADS), lds, s->cur, a);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

// Host-pointer form (what the cv:: signature hands over): copy in, track, copy out, blocking.
extern "C" int rcflow_pyrlk_u8(rc_ctx* ctx, int stream, const uint8_t* prev, size_t prev_step, const uint8_t* next,
                               size_t next_step, int w, int h, const float* prev_pts, float* next_pts, int npts,
                               uint8_t* status, float* err, int win_w, int win_h, int max_level, int crit_type,
                               int max_count, double epsilon, int flags, double min_eig_threshold) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!prev || !next || npts < 0 || (npts > 0 && (!prev_pts || !next_pts || !status)) || w < 1 || h < 1 ||
        prev_step < (size_t)w || next_step < (size_t)w) {
        rc_set_error("bad PyrLK arguments");
        return RC_EINVAL;
    }
    if (npts == 0) return RC_OK;
    RC_HIP(hipSetDevice(ctx->device));
    const size_t img = (size_t)w * h, pts = (size_t)npts * 8;
    int rc = rc_buf_ensure(s->stage_u8, 2 * img);
    if (rc) return rc;
    if ((rc = rc_buf_ensure(s->stage_f32[2], 2 * pts + (size_t)npts * 4 + npts))) return rc;
    uint8_t* d_img = (uint8_t*)s->stage_u8.p;
    char* d = (char*)s->stage_f32[2].p;
    RC_HIP(hipMemcpy2DAsync(d_img, w, prev, prev_step, w, h, hipMemcpyHostToDevice, s->cur));
    RC_HIP(hipMemcpy2DAsync(d_img + img, w, next, next_step, w, h, hipMemcpyHostToDevice, s->cur));
    RC_HIP(hipMemcpyAsync(d, prev_pts, pts, hipMemcpyHostToDevice, s->cur));
    if (flags & 4) RC_HIP(hipMemcpyAsync(d + pts, next_pts, pts, hipMemcpyHostToDevice, s->cur));
    rc = rcflow_pyrlk_dev(ctx, stream, d_img, w, d_img + img, w, w, h, (const float*)d, (float*)(d + pts), npts,
                          (uint8_t*)(d + 2 * pts + (size_t)npts * 4), (float*)(d + 2 * pts), win_w, win_h, max_level,
                          crit_type, max_count, epsilon, flags, min_eig_threshold);
    if (rc) return rc;
    RC_HIP(hipMemcpyAsync(next_pts, d + pts, pts, hipMemcpyDeviceToHost, s->cur));
    if (err) RC_HIP(hipMemcpyAsync(err, d + 2 * pts, (size_t)npts * 4, hipMemcpyDeviceToHost, s->cur));
    RC_HIP(hipMemcpyAsync(status, d + 2 * pts + (size_t)npts * 4, npts, hipMemcpyDeviceToHost, s->cur));
    RC_HIP(hipStreamSynchronize(s->cur));
    return RC_OK;
}
