// exact_kernels.hip -- option "exact": the Farneback stages in the operation order of OpenCV's CPU
// path (modules/video/src/optflow.cpp 4.1.0: FarnebackPolyExp, FarnebackUpdateMatrices,
// FarnebackUpdateFlow_Blur / _GaussianBlur), every float/double operation rounded where the C++
// rounds it (the file is built -ffp-contract=off and writes no fmaf), so that the flow field is the
// CPU path's bit for bit.  Reference call sites: ripcurrents.cpp:215, main.cpp:264, :609, :1119.
//
// Why it exists: the fast kernels reorder sums (fp32 FMA chains on DC-removed data, a Kahan fp32
// solve).  That is within 1e-3 px wherever the 2x2 system is well conditioned, but the
// sigma = 0.3 window of main.cpp:264 is a near-pointwise solve whose determinant
// (r4 r5 - r6^2)^2 vanishes on smooth image regions, and there any rounding difference is amplified
// from scale to scale.  This path removes the rounding differences instead of bounding them.
// It is also the full-size stand-in for the CPU oracle in the GPU tests (a 4K frame pair
// takes milliseconds instead of seconds).
//
// Everything is staged through HBM, one plain kernel per upstream loop:
//   I_k (k_pyr_*, already bit-exact)  ->  k_exact_polyexp  ->  R_k
//   k_exact_flow_init (resize * 1/pyr_scale | zeros)  ->  flow_k
//   k_exact_matrices  ->  M (5 planes)
//   Gaussian window: k_exact_gauss_v -> V (float), k_exact_gauss_h_solve -> flow_k
//   box window:      k_exact_box_vscan -> V (double running column sums, sequential in y like upstream's vsum),
//                    k_exact_box_hscan (running row sums, sequential in x, in place), k_exact_box_solve -> flow_k

#include "rc_device.h"

#define RC_EX_TW 64
#define RC_EX_TH 16

// ------------------------------------------------------------------ FarnebackPolyExp
// Vertical pass (float) for the tile's columns -n .. TW+n-1 (columns clamped = the replicated
// row triplets of upstream), then the horizontal pass with upstream's mix: b1 and b4 take the
// pair sum as a double, every other product is a float product added to a double accumulator.
__global__ __launch_bounds__(256) void k_exact_polyexp(RcPolyArgs a) {
    extern __shared__ __align__(16) float rows[];          // [3][TH][TW + 2n]
    const int n = a.pk.n, CW = RC_EX_TW + 2 * n;
    const int tid = threadIdx.x, z = blockIdx.z;
    const int tx0 = blockIdx.x * RC_EX_TW, ty0 = blockIdx.y * RC_EX_TH;
    const int w = a.w, h = a.h;
    const int slot = (a.slot0 + z * a.zstep) % a.nslots;
    const float* I = a.I + (size_t)slot * a.I_slot_stride;
    float* r0 = rows, *r1 = rows + RC_EX_TH * CW, *r2 = rows + 2 * RC_EX_TH * CW;

    for (int idx = tid; idx < RC_EX_TH * CW; idx += 256) {
        const int ly = idx / CW, lc = idx - ly * CW;
        const int y = min(ty0 + ly, h - 1), c = rc_clampi(tx0 - n + lc, 0, w - 1);
        float t0 = I[(size_t)y * w + c] * a.pk.g[0], t1 = 0.f, t2 = 0.f;
        for (int k = 1; k <= n; k++) {
            const float s0 = I[(size_t)max(y - k, 0) * w + c], s1 = I[(size_t)min(y + k, h - 1) * w + c];
            const float p = s0 + s1;
            t0 = t0 + a.pk.g[k] * p;
            t1 = t1 + a.pk.xg[k] * (s1 - s0);
            t2 = t2 + a.pk.xxg[k] * p;
        }
        r0[idx] = t0; r1[idx] = t1; r2[idx] = t2;
    }
    __syncthreads();
    float4* RA = a.RA + (size_t)slot * a.R_slot_stride;
    float* RB = a.RB + (size_t)slot * a.R_slot_stride;
    for (int idx = tid; idx < RC_EX_TH * RC_EX_TW; idx += 256) {
        const int ly = idx / RC_EX_TW, lx = idx - ly * RC_EX_TW;
        const int x = tx0 + lx, y = ty0 + ly;
        if (x >= w || y >= h) continue;
        const float* q0 = r0 + ly * CW + lx + n, *q1 = r1 + ly * CW + lx + n, *q2 = r2 + ly * CW + lx + n;
        float g0 = a.pk.g[0];
        double b1 = q0[0] * g0, b2 = 0, b3 = q1[0] * g0, b4 = 0, b5 = q2[0] * g0, b6 = 0;
        for (int k = 1; k <= n; k++) {
            const double tg = q0[k] + q0[-k];
            g0 = a.pk.g[k];
            b1 += tg * g0;
            b4 += tg * a.pk.xxg[k];
            b2 += (q0[k] - q0[-k]) * a.pk.xg[k];
            b3 += (q1[k] + q1[-k]) * g0;
            b6 += (q1[k] - q1[-k]) * a.pk.xg[k];
            b5 += (q2[k] + q2[-k]) * g0;
        }
        float4 ra;
        ra.y = (float)(b2 * a.pk.ig11);
        ra.x = (float)(b3 * a.pk.ig11);
        ra.w = (float)(b1 * a.pk.ig03 + b4 * a.pk.ig33);
        ra.z = (float)(b1 * a.pk.ig03 + b5 * a.pk.ig33);
        const size_t p = (size_t)y * w + x;
        RA[p] = ra;
        RB[p] = (float)(b6 * a.pk.ig55);
    }
}

// The same arithmetic for the radii the reference and the OpenCV samples use (15, 7, 5), organised for the
// machine: 64 x 32 outputs per 256-thread block; vertical pass = one thread per (column, 8 rows) over a
// register window of 8 + 2N rows (every input row is loaded once per block instead of once per tap),
// results in three LDS planes; horizontal pass = one thread per (row, 4 pixels) over a register window of
// 4 + 2N columns read as float4s, the six double accumulators of a pixel advanced in upstream's tap order
// (the chains of different accumulators are independent, so plane by plane gives the same bits).
#ifndef RC_EXP_WAVES
#define RC_EXP_WAVES 2
#endif
// TH rows per block: 32, or 8 for launches of a few dozen tiles (a launch of fewer blocks than the GPU holds lasts one
// block's lifetime; the vertical pass's register window spans R + 2 N rows whatever TH is).  Same bits.
template <int N, int TH = 32>
__global__ __launch_bounds__(256, RC_EXP_WAVES) void k_exact_polyexp_t(RcPolyArgs a) {
    constexpr int TW = 64, NP = (N + 3) & ~3, CW = TW + 2 * NP, R = 8, NW = R + 2 * N;
    static_assert(TH % R == 0, "whole groups of R rows");
    constexpr int PX = 2, HW = PX + 2 * NP;                // PX pixels per item; floats under their taps (8-byte aligned reads)
    __shared__ __align__(16) float rows[3][TH][CW];
    const int tid = threadIdx.x, z = blockIdx.z;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int w = a.w, h = a.h;
    const int slot = (a.slot0 + z * a.zstep) % a.nslots;
    const float* I = a.I + (size_t)slot * a.I_slot_stride;

    for (int item = tid; item < CW * (TH / R); item += 256) {
        const int lc = item % CW, grp = item / CW;
        const int c = rc_clampi(tx0 - NP + lc, 0, w - 1), y0 = ty0 + grp * R;
        float win[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) win[j] = I[(size_t)rc_clampi(y0 - N + j, 0, h - 1) * w + c];
#pragma unroll
        for (int o = 0; o < R; o++) {
            float t0 = win[o + N] * a.pk.g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int k = 1; k <= N; k++) {
                const float s0 = win[o + N - k], s1 = win[o + N + k];
                const float p = s0 + s1;
                t0 = t0 + a.pk.g[k] * p;
                t1 = t1 + a.pk.xg[k] * (s1 - s0);
                t2 = t2 + a.pk.xxg[k] * p;
            }
            rows[0][grp * R + o][lc] = t0; rows[1][grp * R + o][lc] = t1; rows[2][grp * R + o][lc] = t2;
        }
    }
    __syncthreads();
    float4* RA = a.RA + (size_t)slot * a.R_slot_stride;
    float* RB = a.RB + (size_t)slot * a.R_slot_stride;
    for (int item = tid; item < TH * (TW / PX); item += 256) {
        const int gp = item % (TW / PX), ly = item / (TW / PX);
        const int y = ty0 + ly, x0 = tx0 + PX * gp;
        if (y >= h || x0 >= w) continue;
        double b1[PX], b2[PX], b3[PX], b4[PX], b5[PX], b6[PX];
        float v[HW];
        // pixel p of the item sits at v[NP + p]; its taps at v[NP + p -+ k]
#define RC_LOAD_PLANE(pl)                                                     \
        _Pragma("unroll") for (int q = 0; q < HW / 2; q++) {                  \
            const float2 t = *(const float2*)&rows[pl][ly][PX * gp + 2 * q];   \
            v[2 * q] = t.x; v[2 * q + 1] = t.y;                               \
        }
        RC_LOAD_PLANE(0)
#pragma unroll
        for (int p = 0; p < PX; p++) {
            const float* q0 = v + NP + p;
            double s1 = q0[0] * a.pk.g[0], s2 = 0, s4 = 0;
#pragma unroll
            for (int k = 1; k <= N; k++) {
                const double tg = q0[k] + q0[-k];
                s1 += tg * a.pk.g[k];
                s4 += tg * a.pk.xxg[k];
                s2 += (q0[k] - q0[-k]) * a.pk.xg[k];
            }
            b1[p] = s1; b2[p] = s2; b4[p] = s4;
        }
        RC_LOAD_PLANE(1)
#pragma unroll
        for (int p = 0; p < PX; p++) {
            const float* q1 = v + NP + p;
            double s3 = q1[0] * a.pk.g[0], s6 = 0;
#pragma unroll
            for (int k = 1; k <= N; k++) {
                s3 += (q1[k] + q1[-k]) * a.pk.g[k];
                s6 += (q1[k] - q1[-k]) * a.pk.xg[k];
            }
            b3[p] = s3; b6[p] = s6;
        }
        RC_LOAD_PLANE(2)
#pragma unroll
        for (int p = 0; p < PX; p++) {
            const float* q2 = v + NP + p;
            double s5 = q2[0] * a.pk.g[0];
#pragma unroll
            for (int k = 1; k <= N; k++) s5 += (q2[k] + q2[-k]) * a.pk.g[k];
            b5[p] = s5;
        }
#undef RC_LOAD_PLANE
#pragma unroll
        for (int p = 0; p < PX; p++) {
            if (x0 + p >= w) break;
            float4 ra;
            ra.y = (float)(b2[p] * a.pk.ig11);
            ra.x = (float)(b3[p] * a.pk.ig11);
            ra.w = (float)(b1[p] * a.pk.ig03 + b4[p] * a.pk.ig33);
            ra.z = (float)(b1[p] * a.pk.ig03 + b5[p] * a.pk.ig33);
            const size_t o = (size_t)y * w + x0 + p;
            RA[o] = ra;
            RB[o] = (float)(b6[p] * a.pk.ig55);
        }
    }
}

template <int N>
static void launch_exact_polyexp_t(const RcPolyArgs& a, int frames, hipStream_t s) {
    const long long tiles32 = (long long)((a.w + 63) / 64) * ((a.h + 31) / 32) * frames;
    // measured in the frame-at-a-time loop at 1080p (main.cpp:264's parameters, one frame per launch): 8-row tiles take the
    // 480 x 270 scale from 16.8 to 11.2 us, leave 960 x 540 at 18.2 and would slow 1920 x 1080 from 42 to 47
    if (tiles32 >= 200) hipLaunchKernelGGL((k_exact_polyexp_t<N, 32>), dim3((a.w + 63) / 64, (a.h + 31) / 32, frames), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_exact_polyexp_t<N, 8>), dim3((a.w + 63) / 64, (a.h + 7) / 8, frames), dim3(256), 0, s, a);
}

void rc_launch_exact_polyexp(const RcPolyArgs& a, int frames, hipStream_t s) {
    if (a.pk.n == 15 && !a.no_fast_u8) { launch_exact_polyexp_t<15>(a, frames, s); return; }
    if (a.pk.n == 7 && !a.no_fast_u8) { launch_exact_polyexp_t<7>(a, frames, s); return; }
    if (a.pk.n == 5 && !a.no_fast_u8) { launch_exact_polyexp_t<5>(a, frames, s); return; }
    const size_t lds = sizeof(float) * 3 * RC_EX_TH * (RC_EX_TW + 2 * a.pk.n);
    dim3 grid((a.w + RC_EX_TW - 1) / RC_EX_TW, (a.h + RC_EX_TH - 1) / RC_EX_TH, frames);
    hipLaunchKernelGGL(k_exact_polyexp, grid, dim3(256), lds, s, a);
}

// ------------------------------------------------------------------ calc(): initial flow of a scale
// zeros at the coarsest scale, else resize(prevFlow, INTER_LINEAR) then *= 1/pyr_scale
__global__ __launch_bounds__(256) void k_exact_flow_init(RcExactArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), z = blockIdx.z;
    if (x >= a.w || y >= a.h) return;
    float2 v = make_float2(0.f, 0.f);
    if (a.fin) {
        const float2* fin = a.fin + (size_t)z * a.fin_pair_stride;
        if (a.fin_w == a.w && a.fin_h == a.h) {
            v = fin[(size_t)y * a.w + x];                  // resize() of an equal size is a copy
        } else {
            float ax, ay;
            const int sx = rc_src_x(x, a.up_scale_x, a.fin_w, ax);
            const int sx1 = min(sx + 1, a.fin_w - 1);
            const int sy = rc_src_y(y, a.up_scale_y, ay);
            const int sy0 = rc_clampi(sy, 0, a.fin_h - 1), sy1 = rc_clampi(sy + 1, 0, a.fin_h - 1);
            const float2* S0 = fin + (size_t)sy0 * a.fin_w;
            const float2* S1 = fin + (size_t)sy1 * a.fin_w;
            const float2 p00 = S0[sx], p01 = S0[sx1], p10 = S1[sx], p11 = S1[sx1];
            const float a0 = 1.f - ax, a1 = ax, b0 = 1.f - ay, b1 = ay;
            const float r0x = p00.x * a0 + p01.x * a1, r1x = p10.x * a0 + p11.x * a1;
            const float r0y = p00.y * a0 + p01.y * a1, r1y = p10.y * a0 + p11.y * a1;
            v.x = r0x * b0 + r1x * b1;
            v.y = r0y * b0 + r1y * b1;
        }
        v.x = v.x * a.up_mul;
        v.y = v.y * a.up_mul;
    }
    a.flow[(size_t)z * a.n + (size_t)y * a.w + x] = v;
}

// ------------------------------------------------------------------ FarnebackUpdateMatrices
// One pixel in optflow.cpp's operation order (no fused multiply-adds): the footprint's position first, then the five
// matrix entries from the pixel's R0, the four R1 texels under the displaced position and the flow.
struct RcExFoot { int x1, y1; float fx, fy; bool inside; };
__device__ __forceinline__ RcExFoot rc_exact_foot(int x, int y, float dx, float dy, int width, int height) {
    RcExFoot f;
    float fx = x + dx, fy = y + dy;
    f.x1 = rc_cvt_i32_x86(floorf(fx)); f.y1 = rc_cvt_i32_x86(floorf(fy));
    f.fx = fx - f.x1;
    f.fy = fy - f.y1;
    f.inside = (unsigned)f.x1 < (unsigned)(width - 1) && (unsigned)f.y1 < (unsigned)(height - 1);
    return f;
}
__device__ __forceinline__ void rc_exact_m5(const RcExFoot& ft, const float4 A0, const float B0, const float4 q00,
                                            const float4 q01, const float4 q10, const float4 q11, float e00, float e01,
                                            float e10, float e11, float dx, float dy, int x, int y, int width,
                                            int height, float (&M)[5]) {
    float r2, r3, r4, r5, r6;
    const float fx = ft.fx, fy = ft.fy;
    if (ft.inside) {
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        r2 = a00 * q00.x + a01 * q01.x + a10 * q10.x + a11 * q11.x;
        r3 = a00 * q00.y + a01 * q01.y + a10 * q10.y + a11 * q11.y;
        r4 = a00 * q00.z + a01 * q01.z + a10 * q10.z + a11 * q11.z;
        r5 = a00 * q00.w + a01 * q01.w + a10 * q10.w + a11 * q11.w;
        r6 = a00 * e00 + a01 * e01 + a10 * e10 + a11 * e11;
        r4 = (A0.z + r4) * 0.5f;
        r5 = (A0.w + r5) * 0.5f;
        r6 = (B0 + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = A0.z;
        r5 = A0.w;
        r6 = B0 * 0.5f;
    }
    r2 = (A0.x - r2) * 0.5f;
    r3 = (A0.y - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    if ((unsigned)(x - 5) >= (unsigned)(width - 10) || (unsigned)(y - 5) >= (unsigned)(height - 10)) {
        const float b0 = 0.14f, b2 = 0.4472f;              // border[] = {.14, .14, .4472, .4472, .4472}
        const int rx = width - x - 1, ry = height - y - 1;
        const float scale = (x < 5 ? (x < 2 ? b0 : b2) : 1.f) * (x >= width - 5 ? (rx < 2 ? b0 : b2) : 1.f) *
                            (y < 5 ? (y < 2 ? b0 : b2) : 1.f) * (y >= height - 5 ? (ry < 2 ? b0 : b2) : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    M[0] = r4 * r4 + r6 * r6;
    M[1] = (r4 + r5) * r6;
    M[2] = r5 * r5 + r6 * r6;
    M[3] = r4 * r2 + r6 * r3;
    M[4] = r6 * r2 + r5 * r3;
}

__global__ __launch_bounds__(256) void k_exact_matrices(RcExactArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), z = blockIdx.z;
    const int width = a.w, height = a.h;
    if (x >= width || y >= height) return;
    const size_t s0 = (size_t)((a.slot0 + z * a.zstep) % a.nslots) * a.n;
    const size_t s1 = (size_t)((a.slot1 + z * a.zstep) % a.nslots) * a.n;
    const float4* RA0 = a.RA + s0; const float* RB0 = a.RB + s0;
    const float4* RA1 = a.RA + s1; const float* RB1 = a.RB + s1;
    const size_t p0 = (size_t)y * width + x;
    const float2 d = a.flow[(size_t)z * a.n + p0];
    const RcExFoot ft = rc_exact_foot(x, y, d.x, d.y, width, height);
    const float4 A0 = RA0[p0];
    const float B0 = RB0[p0];
    float4 q00 = A0, q01 = A0, q10 = A0, q11 = A0;
    float e00 = B0, e01 = B0, e10 = B0, e11 = B0;
    if (ft.inside) {
        const size_t p = (size_t)ft.y1 * width + ft.x1;
        q00 = RA1[p]; q01 = RA1[p + 1]; q10 = RA1[p + width]; q11 = RA1[p + width + 1];
        e00 = RB1[p]; e01 = RB1[p + 1]; e10 = RB1[p + width]; e11 = RB1[p + width + 1];
    }
    float Mv[5];
    rc_exact_m5(ft, A0, B0, q00, q01, q10, q11, e00, e01, e10, e11, d.x, d.y, x, y, width, height, Mv);
    float* M = a.M + (size_t)z * 5 * a.n + p0;
    M[0] = Mv[0];
    M[a.n] = Mv[1];
    M[2 * a.n] = Mv[2];
    M[3 * a.n] = Mv[3];
    M[4 * a.n] = Mv[4];
}

__device__ __forceinline__ float2 rc_exact_solve(double g11, double g12, double g22, double h1, double h2) {
    const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
    float2 f;
    f.x = (float)((g11 * h2 - g12 * h1) * idet);
    f.y = (float)((g22 * h1 - g12 * h2) * idet);
    return f;
}

// ------------------------------------------------------------------ FarnebackUpdateFlow_GaussianBlur
// vertical pass: vsum = srow[m] * k[0]; vsum += (srow[m+i] + srow[m-i]) * k[i]   (float)
__global__ __launch_bounds__(256) void k_exact_gauss_v(RcExactArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int z = blockIdx.z / 5, c = blockIdx.z - z * 5;
    if (x >= a.w || y >= a.h) return;
    const float* M = a.M + ((size_t)z * 5 + c) * a.n;
    float s0 = M[(size_t)y * a.w + x] * a.win.k[0];
    for (int i = 1; i <= a.win.m; i++)
        s0 += (M[(size_t)min(y + i, a.h - 1) * a.w + x] + M[(size_t)max(y - i, 0) * a.w + x]) * a.win.k[i];
    ((float*)a.V)[((size_t)z * 5 + c) * a.n + (size_t)y * a.w + x] = s0;
}
// horizontal pass (replicated columns) + solve in double
__global__ __launch_bounds__(256) void k_exact_gauss_h_solve(RcExactArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), z = blockIdx.z;
    if (x >= a.w || y >= a.h) return;
    float hs[5];
    for (int c = 0; c < 5; c++) {
        const float* V = (const float*)a.V + ((size_t)z * 5 + c) * a.n + (size_t)y * a.w;
        float sum = V[x] * a.win.k[0];
        for (int i = 1; i <= a.win.m; i++) sum += a.win.k[i] * (V[max(x - i, 0)] + V[min(x + i, a.w - 1)]);
        hs[c] = sum;
    }
    const float2 f = rc_exact_solve(hs[0], hs[1], hs[2], hs[3], hs[4]);
    if (a.out) *(float2*)(a.out + (size_t)z * a.out_pair_stride + (size_t)y * a.out_step + (size_t)x * 8) = f;
    else a.flow[(size_t)z * a.n + (size_t)y * a.w + x] = f;
}

// ------------------------------------------------------------------ FarnebackUpdateFlow_Blur
// upstream's running column sums: double accumulators fed with FLOAT differences of rows, so the
// rounding of every (srow1 - srow0) is carried down the column; replayed sequentially per column.
// The loads do not depend on the running sum: eight rows of them are issued ahead of the eight dependent additions.
__global__ __launch_bounds__(64) void k_exact_box_vscan(RcExactArgs a) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int z = blockIdx.y / 5, c = blockIdx.y - z * 5;
    if (x >= a.w) return;
    const int m = a.win.m, h = a.h, w = a.w;
    const float* M = a.M + ((size_t)z * 5 + c) * a.n + x;
    double* V = (double*)a.V + ((size_t)z * 5 + c) * a.n + x;
    double vsum = M[0] * (m + 2);                          // float product
    for (int y = 1; y < m; y++) vsum += M[(size_t)min(y, h - 1) * w];
    constexpr int U = 8;
    for (int y0 = 0; y0 < h; y0 += U) {
        float d[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int y = min(y0 + u, h - 1);
            d[u] = M[(size_t)min(y + m, h - 1) * w] - M[(size_t)max(y - m - 1, 0) * w];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (y0 + u < h) {
                vsum += d[u];
                V[(size_t)(y0 + u) * w] = vsum;
            }
        }
    }
}
// running row sums (double), one thread per (row, channel): G(x) = G(x - 1) + V(x + m) - V(x - m - 1), replicated
// borders; eight steps of loads ahead of the eight dependent additions.  (Moving V in and G out through LDS tiles so
// that global memory sees whole row segments was measured slower with one wave per block: 822 vs 544 us per 1080p pair.)
__global__ __launch_bounds__(64) void k_exact_box_hscan(RcExactArgs a) {
    const int y = blockIdx.x * 64 + threadIdx.x;
    const int z = blockIdx.y / 5, c = blockIdx.y - z * 5;
    if (y >= a.h) return;
    const int m = a.win.m, w = a.w;
    const double* V = (const double*)a.V + ((size_t)z * 5 + c) * a.n + (size_t)y * w;
    double* G = (double*)a.G + ((size_t)z * 5 + c) * a.n + (size_t)y * w;
    double g = V[0] * (m + 2);
    for (int x = 1; x < m; x++) g += V[min(x, w - 1)];
    constexpr int U = 8;
    for (int x0 = 0; x0 < w; x0 += U) {
        double vn[U], vo[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int x = min(x0 + u, w - 1);
            vn[u] = V[min(x + m, w - 1)];
            vo[u] = V[max(x - m - 1, 0)];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (x0 + u < w) {
                g += vn[u] - vo[u];
                G[x0 + u] = g;
            }
        }
    }
}
// scale + solve, one thread per pixel
__global__ __launch_bounds__(256) void k_exact_box_solve(RcExactArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), z = blockIdx.z;
    if (x >= a.w || y >= a.h) return;
    const double* G = (const double*)a.G + (size_t)z * 5 * a.n + (size_t)y * a.w + x;
    const double scale = a.win.box_scale;
    const float2 f = rc_exact_solve(G[0] * scale, G[a.n] * scale, G[2 * a.n] * scale, G[3 * a.n] * scale, G[4 * a.n] * scale);
    if (a.out) *(float2*)(a.out + (size_t)z * a.out_pair_stride + (size_t)y * a.out_step + (size_t)x * 8) = f;
    else a.flow[(size_t)z * a.n + (size_t)y * a.w + x] = f;
}

// ---- box windows of radius 1 and 2 (winsize 3: ripcurrents.cpp:215; winsize 5: the Android fork): the same two
// sequential scans with every global access coalesced.  The column scan hands its running sums over TRANSPOSED
// (V^T[channel][x][y], rows padded to a multiple of 16), through a 64 x 16 LDS tile per wave, so that the row scan --
// one lane per image row -- reads 512 contiguous bytes per step.  The row scan keeps all five channels of its row,
// solves the 2 x 2 system in place (G never goes to memory) and writes the flow through a second LDS transposition
// as whole 128-byte row segments.  The next sixteen steps' column sums are loaded while the current sixteen are
// consumed; what a step subtracts is what an earlier step added, kept in registers.
template <int MM>
__global__ __launch_bounds__(64) void k_exact_box_vscan_t(RcExactArgs a, int hp) {
    __shared__ double T[64 * 17];
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 64, x = min(x0 + lane, a.w - 1);
    const int z = blockIdx.y / 5, c = blockIdx.y - z * 5;
    constexpr int m = MM;
    const int h = a.h, w = a.w;
    const float* M = a.M + ((size_t)z * 5 + c) * a.n + x;
    double* VT = (double*)a.V + ((size_t)z * 5 + c) * (size_t)w * hp;
    double vsum = M[0] * (m + 2);                          // float product
    for (int y = 1; y < m; y++) vsum += M[(size_t)min(y, h - 1) * w];
    for (int y0 = 0; y0 < h; y0 += 16) {
        float d[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int y = min(y0 + u, h - 1);
            d[u] = M[(size_t)min(y + m, h - 1) * w] - M[(size_t)max(y - m - 1, 0) * w];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            vsum += d[u];                                  // (rows past the image: values nobody reads)
            T[lane * 17 + u] = vsum;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int xl = i * 8 + (lane >> 3), yy = (lane & 7) * 2;
            const double v0 = T[xl * 17 + yy], v1 = T[xl * 17 + yy + 1];
            if (x0 + xl < w) *(double2*)(VT + (size_t)(x0 + xl) * hp + y0 + yy) = make_double2(v0, v1);
        }
        __syncthreads();
    }
}

// (`ablate` 16; measured slower than the two kernels it replaces: 266 vs 258 us per 1080p pair in 32-pair clips, 519 vs 357 in
// 8-pair clips -- one wave per 64 columns and pair is too little parallelism for its chains of loads.)
// FarnebackUpdateMatrices evaluated inside the column scan: M never goes to memory.  One lane per image column walks down
// the rows; the matrices of the row entering the window are computed on the spot (the rows leaving it wait in a register
// ring), four rows per pipeline stage: the stage after next loads its flow and R0 texels, the next stage has its R1
// gathers in flight, the current one is evaluated.  Same arithmetic per pixel as k_exact_matrices + k_exact_box_vscan_t.
template <int MM>
__global__ __launch_bounds__(64) void k_exact_box_mvscan_t(RcExactArgs a, int hp) {
    constexpr int m = MM, L = 2 * MM + 1, S = 4, FL = 8;
    __shared__ double T[5][64 * (FL + 1)];
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 64, x = min(x0 + lane, a.w - 1);
    const int z = blockIdx.y;
    const int h = a.h, w = a.w;
    const size_t s0 = (size_t)((a.slot0 + z * a.zstep) % a.nslots) * a.n;
    const size_t s1 = (size_t)((a.slot1 + z * a.zstep) % a.nslots) * a.n;
    const float4* __restrict__ RA0 = a.RA + s0; const float* __restrict__ RB0 = a.RB + s0;
    const float4* __restrict__ RA1 = a.RA + s1; const float* __restrict__ RB1 = a.RB + s1;
    const float2* __restrict__ flow = a.flow + (size_t)z * a.n;
    double* VT = (double*)a.V + (size_t)z * 5 * (size_t)w * hp;
    const size_t cs = (size_t)w * hp;

    struct St1 { float2 f; float4 A0; float B0; };
    struct Ga { float4 q00, q01, q10, q11; float e00, e01, e10, e11; };
    auto load1 = [&](int r, St1& st) {
        const size_t p0 = (size_t)r * w + x;
        st.f = flow[p0]; st.A0 = RA0[p0]; st.B0 = RB0[p0];
    };
    auto gather = [&](int r, const St1& st, Ga& g) {
        const RcExFoot ft = rc_exact_foot(x, r, st.f.x, st.f.y, w, h);
        // (a footprint outside the image is not used: its four loads read the pixel's own texel)
        const size_t p = ft.inside ? (size_t)ft.y1 * w + ft.x1 : (size_t)r * w + x;
        const size_t pw = ft.inside ? (size_t)w : 0, p1 = ft.inside ? 1 : 0;
        g.q00 = RA1[p]; g.q01 = RA1[p + p1]; g.q10 = RA1[p + pw]; g.q11 = RA1[p + pw + p1];
        g.e00 = RB1[p]; g.e01 = RB1[p + p1]; g.e10 = RB1[p + pw]; g.e11 = RB1[p + pw + p1];
    };
    auto mats = [&](int r, const St1& st, const Ga& g, float (&Mv)[5]) {
        const RcExFoot ft = rc_exact_foot(x, r, st.f.x, st.f.y, w, h);
        rc_exact_m5(ft, st.A0, st.B0, g.q00, g.q01, g.q10, g.q11, g.e00, g.e01, g.e10, g.e11, st.f.x, st.f.y, x, r, w, h, Mv);
    };

    // rows 0 .. m-1: upstream's initial column sums, and the ring = M of rows -m-1 .. m-1 (clamped)
    float Mi[MM][5];
#pragma unroll
    for (int j = 0; j < MM; j++) {
        St1 st; Ga g;
        const int r = min(j, h - 1);
        load1(r, st); gather(r, st, g); mats(r, st, g, Mi[j]);
    }
    double vsum[5];
    float ring[L][5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        vsum[c] = Mi[0][c] * (m + 2);                        // float product
#pragma unroll
        for (int y = 1; y < m; y++) vsum[c] += Mi[y][c];
#pragma unroll
        for (int j = 0; j < L; j++) ring[j][c] = Mi[j - m - 1 > 0 ? j - m - 1 : 0][c];
    }

    St1 sa[S], sb[S], sc[S];
    Ga ga[S], gb[S];
    auto row_of = [&](int y) { return min(y + m, h - 1); };   // the row entering the window at step y
#pragma unroll
    for (int u = 0; u < S; u++) load1(row_of(u), sa[u]);
#pragma unroll
    for (int u = 0; u < S; u++) load1(row_of(S + u), sb[u]);
#pragma unroll
    for (int u = 0; u < S; u++) gather(row_of(u), sa[u], ga[u]);
    for (int y0 = 0; y0 < hp; y0 += S) {
#pragma unroll
        for (int u = 0; u < S; u++) load1(row_of(y0 + 2 * S + u), sc[u]);
#pragma unroll
        for (int u = 0; u < S; u++) gather(row_of(y0 + S + u), sb[u], gb[u]);
#pragma unroll
        for (int u = 0; u < S; u++) {
            float Mn[5];
            mats(row_of(y0 + u), sa[u], ga[u], Mn);
#pragma unroll
            for (int c = 0; c < 5; c++) {
                vsum[c] += Mn[c] - ring[0][c];               // float difference, double sum (rows past the image: nobody reads them)
#pragma unroll
                for (int j = 0; j < L - 1; j++) ring[j][c] = ring[j + 1][c];
                ring[L - 1][c] = Mn[c];
                T[c][lane * (FL + 1) + ((y0 + u) & (FL - 1))] = vsum[c];
            }
        }
        if (((y0 + S) & (FL - 1)) == 0) {
            const int yb = y0 + S - FL;
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 5; c++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int xl = i * 16 + (lane >> 2), yy = (lane & 3) * 2;
                    const double v0 = T[c][xl * (FL + 1) + yy], v1 = T[c][xl * (FL + 1) + yy + 1];
                    if (x0 + xl < w) *(double2*)(VT + c * cs + (size_t)(x0 + xl) * hp + yb + yy) = make_double2(v0, v1);
                }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < S; u++) { sa[u] = sb[u]; sb[u] = sc[u]; ga[u] = gb[u]; }
    }
}

template <int MM>
__global__ __launch_bounds__(64) void k_exact_box_hsolve_t(RcExactArgs a, int hp) {
    constexpr int m = MM, L = 2 * MM + 1, U = 16;
    static_assert(L <= U, "the ring is refilled from one block of steps");
    __shared__ float2 T[64 * 17];
    const int lane = threadIdx.x;
    const int y0 = blockIdx.x * 64, y = min(y0 + lane, a.h - 1);
    const int z = blockIdx.y;
    const int w = a.w;
    const double scale = a.win.box_scale;
    const double* VT = (const double*)a.V + (size_t)z * 5 * (size_t)w * hp + y;
    const size_t cs = (size_t)w * hp;                      // channel stride
    // ring[c][j] = V(x - m - 1 + j) before step x, j = 0 .. 2m  (clamped columns); g = upstream's initial sums
    double ring[5][L], g[5];
#pragma unroll
    for (int c = 0; c < 5; c++) {
#pragma unroll
        for (int j = 0; j < L; j++) ring[c][j] = VT[c * cs + (size_t)rc_clampi(j - m - 1, 0, w - 1) * hp];
        const double v0 = VT[c * cs];
        g[c] = v0 * (m + 2);
        for (int xx = 1; xx < m; xx++) g[c] += VT[c * cs + (size_t)min(xx, w - 1) * hp];
    }
    double nv[2][U][5];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int c = 0; c < 5; c++) nv[0][u][c] = VT[c * cs + (size_t)min(u + m, w - 1) * hp];
    char* outp;
    size_t ostep;
    if (a.out) { outp = a.out + (size_t)z * a.out_pair_stride; ostep = a.out_step; }
    else { outp = (char*)(a.flow + (size_t)z * a.n); ostep = (size_t)w * 8; }

    auto block = [&](const double (&cur)[U][5], double (&nxt)[U][5], int xb) {
        // the next block's column sums, in flight while this block is consumed
        if (xb + U < w) {
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int c = 0; c < 5; c++) nxt[u][c] = VT[c * cs + (size_t)min(xb + U + u + m, w - 1) * hp];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            double gs[5];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const double vold = u >= L ? cur[u - L][c] : ring[c][u];
                g[c] += cur[u][c] - vold;
                gs[c] = g[c] * scale;
            }
            T[lane * 17 + u] = rc_exact_solve(gs[0], gs[1], gs[2], gs[3], gs[4]);
        }
#pragma unroll
        for (int c = 0; c < 5; c++)
#pragma unroll
            for (int j = 0; j < L; j++) ring[c][j] = cur[U - L + j][c];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int yl = i * 8 + (lane >> 3), xx = (lane & 7) * 2;
            const float2 f0 = T[yl * 17 + xx], f1 = T[yl * 17 + xx + 1];
            char* o = outp + (size_t)(y0 + yl) * ostep + (size_t)(xb + xx) * 8;
            if (y0 + yl < a.h) {
                if (xb + xx + 1 < w) *(float4*)o = make_float4(f0.x, f0.y, f1.x, f1.y);
                else if (xb + xx < w) *(float2*)o = f0;
            }
        }
        __syncthreads();
    };
    for (int xb = 0; xb < w; xb += 2 * U) {
        block(nv[0], nv[1], xb);
        if (xb + U < w) block(nv[1], nv[0], xb + U);
    }
}

// ------------------------------------------------------------------ launchers
void rc_launch_exact_flow_init(const RcExactArgs& a, int pairs, hipStream_t s) {
    hipLaunchKernelGGL(k_exact_flow_init, dim3((a.w + 63) / 64, (a.h + 3) / 4, pairs), dim3(256), 0, s, a);
}
void rc_launch_exact_matrices(const RcExactArgs& a, int pairs, hipStream_t s) {
    hipLaunchKernelGGL(k_exact_matrices, dim3((a.w + 63) / 64, (a.h + 3) / 4, pairs), dim3(256), 0, s, a);
}
// Box windows of winsize 3 / 5 on 16-byte aligned flow rows: matrices + column scan in one kernel, then row scan + solve
static bool rc_exact_box_t_ok(const RcExactArgs& a) {
    // (the flow rows are written as float4 pairs: 16-byte aligned rows only; any other shape takes the plain scans)
    const bool al16 = a.out ? ((((size_t)a.out) | a.out_step | a.out_pair_stride) & 15) == 0 : (a.w % 2) == 0;
    return !a.win.gaussian && (a.win.m == 1 || a.win.m == 2) && al16 && !a.plain_scans;
}
int rc_exact_iteration_fused_ok(const RcExactArgs& a) { return rc_exact_box_t_ok(a) && a.fused_matrices; }
void rc_launch_exact_iteration_fused(const RcExactArgs& a, int pairs, hipStream_t s) {
    const int hp = (a.h + 15) & ~15;
    if (a.win.m == 1) {
        hipLaunchKernelGGL(k_exact_box_mvscan_t<1>, dim3((a.w + 63) / 64, pairs), dim3(64), 0, s, a, hp);
        hipLaunchKernelGGL(k_exact_box_hsolve_t<1>, dim3((a.h + 63) / 64, pairs), dim3(64), 0, s, a, hp);
    } else {
        hipLaunchKernelGGL(k_exact_box_mvscan_t<2>, dim3((a.w + 63) / 64, pairs), dim3(64), 0, s, a, hp);
        hipLaunchKernelGGL(k_exact_box_hsolve_t<2>, dim3((a.h + 63) / 64, pairs), dim3(64), 0, s, a, hp);
    }
}
void rc_launch_exact_window_solve(const RcExactArgs& a, int pairs, hipStream_t s) {
    if (a.win.gaussian) {
        hipLaunchKernelGGL(k_exact_gauss_v, dim3((a.w + 63) / 64, (a.h + 3) / 4, pairs * 5), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_exact_gauss_h_solve, dim3((a.w + 63) / 64, (a.h + 3) / 4, pairs), dim3(256), 0, s, a);
    } else {
        const int hp = (a.h + 15) & ~15;
        if (rc_exact_box_t_ok(a)) {
            if (a.win.m == 1) {
                hipLaunchKernelGGL(k_exact_box_vscan_t<1>, dim3((a.w + 63) / 64, pairs * 5), dim3(64), 0, s, a, hp);
                hipLaunchKernelGGL(k_exact_box_hsolve_t<1>, dim3((a.h + 63) / 64, pairs), dim3(64), 0, s, a, hp);
            } else {
                hipLaunchKernelGGL(k_exact_box_vscan_t<2>, dim3((a.w + 63) / 64, pairs * 5), dim3(64), 0, s, a, hp);
                hipLaunchKernelGGL(k_exact_box_hsolve_t<2>, dim3((a.h + 63) / 64, pairs), dim3(64), 0, s, a, hp);
            }
            return;
        }
        hipLaunchKernelGGL(k_exact_box_vscan, dim3((a.w + 63) / 64, pairs * 5), dim3(64), 0, s, a);
        hipLaunchKernelGGL(k_exact_box_hscan, dim3((a.h + 63) / 64, pairs * 5), dim3(64), 0, s, a);
        hipLaunchKernelGGL(k_exact_box_solve, dim3((a.w + 63) / 64, (a.h + 3) / 4, pairs), dim3(256), 0, s, a);
    }
}
