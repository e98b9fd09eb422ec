// farneback_kernels.hip -- hand-written gfx950 kernels for dense Farneback optical flow.
//
// Replaces the arithmetic behind cv::calcOpticalFlowFarneback as the reference calls it
// (RipCurrents_main/ripcurrents.cpp:215; main.cpp:264,609,742,961,1119,1481).  Stages
// follow SURVEY.md section 8(a):
//   k_pyr_level   A1  convertTo(32F) + GaussianBlur(full res, REFLECT_101) + resize(LINEAR)
//   k_polyexp     A2  FarnebackPolyExp            -> R = (y, x, yy, xx | xy) planes
//   k_flow_iter   A3+A4/A5+A6  FarnebackUpdateMatrices + window blur + 2x2 solve, fused:
//                 the 5-channel matrix image M never goes to HBM (it lives in LDS)
// Data layout in HBM (all fp32):
//   I_k   [slot][h][w]            float
//   RA_k  [slot][h][w]            float4 (y, x, yy, xx)  -- one 16-B load per pixel/texel
//   RB_k  [slot][h][w]            float  (xy)
//   flow  [pair][h][w]            float2 (x, y)  == CV_32FC2
// Compiled with -ffp-contract=off; fused multiply-adds are written explicitly (RC_FMA)
// in the convolution loops and nowhere else, so the gather / matrix / resize arithmetic
// rounds operation by operation like the scalar C++ it has to match.

#include "rc_common.h"

#define RC_BLOCK 256

// ---------------------------------------------------------------------------------
__device__ __forceinline__ int rc_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
__device__ __forceinline__ int rc_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// resize.cpp INTER_LINEAR source column for destination dx (alpha = (1-ax, ax))
__device__ __forceinline__ int rc_src_x(int dx, double scale_x, int sw, float& ax) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
    ax = fx;
    return sx;
}
// source row (unclamped) for destination dy; rows sy and sy+1 are clamped by the caller
__device__ __forceinline__ int rc_src_y(int dy, double scale_y, float& ay) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    ay = fy - sy;
    return sy;
}

// XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give
// each XCD one contiguous run of tiles; neighbouring tiles then share halo lines in
// that XCD's L2.  Bijective for any tile count.
__device__ __forceinline__ int rc_xcd_remap(int b, int nt) {
    int q = nt >> 3, r = nt & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ===================================================================== A1 pyramid
// One block = tw x th output pixels of scale k.  The 8-bit source region the tile needs
// (blur radius + resize footprint) is staged in LDS, the horizontal blur is evaluated
// only at the two source columns each output column samples, then each thread does the
// vertical blur at its 2x2 sample points and the bilinear resize.
__global__ __launch_bounds__(RC_BLOCK) void k_pyr_level(RcPyrArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* reg = smem;
    float* rp = (float*)(smem + (size_t)a.reg_hmax * a.reg_wp);   // [reg_h][2*tw]
    const int tid = threadIdx.x;
    const int z = blockIdx.z;
    const int tx0 = blockIdx.x * a.tw, ty0 = blockIdx.y * a.th;
    const int r = a.ksize >> 1;
    const int W0 = a.W0, H0 = a.H0;
    float dummy;

    const int dxb = min(tx0 + a.tw, a.w) - 1, dyb = min(ty0 + a.th, a.h) - 1;
    const int reg_x0 = rc_src_x(tx0, a.scale_x, W0, dummy) - r;
    const int reg_x1 = min(rc_src_x(dxb, a.scale_x, W0, dummy) + 1, W0 - 1) + r;
    const int reg_y0 = rc_clampi(rc_src_y(ty0, a.scale_y, dummy), 0, H0 - 1) - r;
    const int reg_y1 = rc_clampi(rc_src_y(dyb, a.scale_y, dummy) + 1, 0, H0 - 1) + r;
    const int reg_w = reg_x1 - reg_x0 + 1, reg_h = reg_y1 - reg_y0 + 1;

    const uint8_t* src = a.src + (size_t)z * a.src_frame_stride;
    for (int idx = tid; idx < reg_h * reg_w; idx += RC_BLOCK) {
        int i = idx / reg_w, j = idx - i * reg_w;
        int sy = rc_reflect101(reg_y0 + i, H0), sx = rc_reflect101(reg_x0 + j, W0);
        reg[i * a.reg_wp + j] = src[(size_t)sy * a.src_step + sx];
    }
    __syncthreads();

    // horizontal blur (RowFilter order of smooth.cpp; SymmRowSmallFilter for ksize<=5)
    const int tw2 = 2 * a.tw;
    const float* kern = a.kern;
    for (int idx = tid; idx < reg_h * tw2; idx += RC_BLOCK) {
        int i = idx / tw2, j = idx - i * tw2;
        int dx = min(tx0 + (j >> 1), a.w - 1);
        int sx = rc_src_x(dx, a.scale_x, W0, dummy);
        if (j & 1) sx = min(sx + 1, W0 - 1);
        const unsigned char* S = reg + i * a.reg_wp + (sx - reg_x0);
        float s0;
        if (a.ksize == 3) {
            s0 = (float)S[0] * kern[1] + ((float)S[-1] + (float)S[1]) * kern[2];
        } else if (a.ksize == 5) {
            s0 = (float)S[0] * kern[2] + ((float)S[-1] + (float)S[1]) * kern[3] +
                 ((float)S[-2] + (float)S[2]) * kern[4];
        } else {
            s0 = kern[0] * (float)S[-r];
            for (int k = 1; k < a.ksize; k++) s0 += kern[k] * (float)S[k - r];
        }
        rp[idx] = s0;
    }
    __syncthreads();

    const int lx = tid % a.tw, ly = tid / a.tw;
    const int dx = tx0 + lx, dy = ty0 + ly;
    if (ly < a.th && dx < a.w && dy < a.h) {
        float ax, ay;
        rc_src_x(dx, a.scale_x, W0, ax);
        int sy = rc_src_y(dy, a.scale_y, ay);
        int i0 = rc_clampi(sy, 0, H0 - 1) - reg_y0, i1 = rc_clampi(sy + 1, 0, H0 - 1) - reg_y0;
        const float* c0 = rp + 2 * lx;
        // vertical blur (SymmColumnFilter order) at the four sample points
        float b00 = kern[r] * c0[i0 * tw2], b01 = kern[r] * c0[i0 * tw2 + 1];
        float b10 = kern[r] * c0[i1 * tw2], b11 = kern[r] * c0[i1 * tw2 + 1];
        for (int k = 1; k <= r; k++) {
            float kk = kern[r + k];
            b00 += kk * (c0[(i0 + k) * tw2] + c0[(i0 - k) * tw2]);
            b01 += kk * (c0[(i0 + k) * tw2 + 1] + c0[(i0 - k) * tw2 + 1]);
            b10 += kk * (c0[(i1 + k) * tw2] + c0[(i1 - k) * tw2]);
            b11 += kk * (c0[(i1 + k) * tw2 + 1] + c0[(i1 - k) * tw2 + 1]);
        }
        float a0 = 1.f - ax, a1 = ax, w0 = 1.f - ay, w1 = ay;
        float r0 = b00 * a0 + b01 * a1;
        float r1 = b10 * a0 + b11 * a1;
        int slot = (a.dslot0 + z) % a.nslots;
        a.dst[(size_t)slot * a.dst_slot_stride + (size_t)dy * a.w + dx] = r0 * w0 + r1 * w1;
    }
}

void rc_launch_pyr(const RcPyrArgs& a, int frames, size_t lds, hipStream_t s) {
    dim3 grid((a.w + a.tw - 1) / a.tw, (a.h + a.th - 1) / a.th, frames);
    hipLaunchKernelGGL(k_pyr_level, grid, dim3(RC_BLOCK), lds, s, a);
}

// ===================================================================== A2 polyexp
// 64x32 output tile per 256-thread block.  Separable: horizontal pass first (three sums
// per pixel: g, x*g, x*x*g), staged through LDS, vertical pass last so that every lane of
// a wave owns one column and all LDS reads and the 16-B global stores are conflict-free
// and coalesced.  Each thread of the vertical pass produces 8 rows from a register
// window.  A per-tile constant (the tile-centre pixel) is subtracted before the sums and
// its exact contribution (pk.kdc) is added back in the double-precision epilogue: the
// yy/xx coefficients are differences of O(100) sums, and this keeps them at fp32's best.
template <int R>
__global__ __launch_bounds__(RC_BLOCK) void k_polyexp(RcPolyArgs a) {
    constexpr int TW = 64, TH = 32, RP = (R + 3) & ~3;
    constexpr int INW = TW + 2 * RP, INH = TH + 2 * R;
    constexpr int NV = 4 + 2 * RP;
    extern __shared__ __align__(16) float smf[];
    float* tin = smf;                // [INH][INW]
    float* hs = smf + INH * INW;     // [3][INH][TW]
    const int tid = threadIdx.x;
    const int z = blockIdx.z;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int w = a.w, h = a.h;
    const int slot = (a.slot0 + z) % a.nslots;
    const float* I = a.I + (size_t)slot * a.I_slot_stride;

    const float dc = I[(size_t)min(ty0 + TH / 2, h - 1) * w + min(tx0 + TW / 2, w - 1)];

    for (int idx = tid; idx < INH * INW; idx += RC_BLOCK) {
        int i = idx / INW, j = idx - i * INW;
        int gy = rc_clampi(ty0 - R + i, 0, h - 1), gx = rc_clampi(tx0 - RP + j, 0, w - 1);
        tin[idx] = I[(size_t)gy * w + gx] - dc;
    }
    __syncthreads();

    // horizontal pass: item = (row i, group of 4 pixels)
    for (int idx = tid; idx < INH * (TW / 4); idx += RC_BLOCK) {
        int i = idx / (TW / 4), g4 = idx - i * (TW / 4);
        float v[NV];
        const float4* p4 = (const float4*)(tin + i * INW + 4 * g4);
#pragma unroll
        for (int q = 0; q < NV / 4; q++) {
            float4 t = p4[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        float h0[4], h1[4], h2[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int c = RP + p;
            float s0 = v[c] * a.pk.g[0], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 1; k <= R; k++) {
                float sm = v[c + k] + v[c - k], df = v[c + k] - v[c - k];
                s0 = RC_FMA(sm, a.pk.g[k], s0);
                s1 = RC_FMA(df, a.pk.xg[k], s1);
                s2 = RC_FMA(sm, a.pk.xxg[k], s2);
            }
            h0[p] = s0; h1[p] = s1; h2[p] = s2;
        }
        float4* o = (float4*)(hs + i * TW + 4 * g4);
        o[0] = make_float4(h0[0], h0[1], h0[2], h0[3]);
        o[INH * TW / 4] = make_float4(h1[0], h1[1], h1[2], h1[3]);
        o[2 * INH * TW / 4] = make_float4(h2[0], h2[1], h2[2], h2[3]);
    }
    __syncthreads();

    // vertical pass: lane = column, 8 output rows per thread
    const int x = tid & 63, o0 = (tid >> 6) * 8;
    constexpr int NW = 8 + 2 * R;
    float b1[8], b2[8], b3[8], b4[8], b5[8], b6[8];
    {
        float c[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) c[q] = hs[(o0 + q) * TW + x];
#pragma unroll
        for (int o = 0; o < 8; o++) {
            float s1 = c[o + R] * a.pk.g[0], s3 = 0.f, s5 = 0.f;
#pragma unroll
            for (int k = 1; k <= R; k++) {
                float sm = c[o + R + k] + c[o + R - k], df = c[o + R + k] - c[o + R - k];
                s1 = RC_FMA(sm, a.pk.g[k], s1);
                s3 = RC_FMA(df, a.pk.xg[k], s3);
                s5 = RC_FMA(sm, a.pk.xxg[k], s5);
            }
            b1[o] = s1; b3[o] = s3; b5[o] = s5;
        }
    }
    {
        float c[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) c[q] = hs[INH * TW + (o0 + q) * TW + x];
#pragma unroll
        for (int o = 0; o < 8; o++) {
            float s2 = c[o + R] * a.pk.g[0], s6 = 0.f;
#pragma unroll
            for (int k = 1; k <= R; k++) {
                s2 = RC_FMA(c[o + R + k] + c[o + R - k], a.pk.g[k], s2);
                s6 = RC_FMA(c[o + R + k] - c[o + R - k], a.pk.xg[k], s6);
            }
            b2[o] = s2; b6[o] = s6;
        }
    }
    {
        float c[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) c[q] = hs[2 * INH * TW + (o0 + q) * TW + x];
#pragma unroll
        for (int o = 0; o < 8; o++) {
            float s4 = c[o + R] * a.pk.g[0];
#pragma unroll
            for (int k = 1; k <= R; k++) s4 = RC_FMA(c[o + R + k] + c[o + R - k], a.pk.g[k], s4);
            b4[o] = s4;
        }
    }
    const int gx = tx0 + x;
    if (gx < w) {
        float4* RA = a.RA + (size_t)slot * a.R_slot_stride;
        float* RB = a.RB + (size_t)slot * a.R_slot_stride;
        const double dck = (double)dc * a.pk.kdc;
#pragma unroll
        for (int o = 0; o < 8; o++) {
            int gy = ty0 + o0 + o;
            if (gy < h) {
                float4 ra;
                ra.x = (float)((double)b3[o] * a.pk.ig11);
                ra.y = (float)((double)b2[o] * a.pk.ig11);
                ra.z = (float)((double)b1[o] * a.pk.ig03 + (double)b5[o] * a.pk.ig33 + dck);
                ra.w = (float)((double)b1[o] * a.pk.ig03 + (double)b4[o] * a.pk.ig33 + dck);
                size_t p = (size_t)gy * w + gx;
                RA[p] = ra;
                RB[p] = (float)((double)b6[o] * a.pk.ig55);
            }
        }
    }
}

template <int R>
static void launch_polyexp_t(const RcPolyArgs& a, int frames, hipStream_t s) {
    constexpr int RP = (R + 3) & ~3;
    size_t lds = sizeof(float) * ((size_t)(32 + 2 * R) * (64 + 2 * RP) + 3 * (size_t)(32 + 2 * R) * 64);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_polyexp<R>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr_set = true;
    }
    dim3 grid((a.w + 63) / 64, (a.h + 31) / 32, frames);
    hipLaunchKernelGGL(k_polyexp<R>, grid, dim3(RC_BLOCK), lds, s, a);
}

void rc_launch_polyexp(const RcPolyArgs& a, int frames, hipStream_t s) {
    int n = a.pk.n_eff;
    if (n <= 3) launch_polyexp_t<3>(a, frames, s);
    else if (n <= 5) launch_polyexp_t<5>(a, frames, s);
    else if (n <= 7) launch_polyexp_t<7>(a, frames, s);
    else if (n <= 9) launch_polyexp_t<9>(a, frames, s);
    else if (n <= 12) launch_polyexp_t<12>(a, frames, s);
    else if (n <= 16) launch_polyexp_t<16>(a, frames, s);
    else if (n <= 24) launch_polyexp_t<24>(a, frames, s);
    else launch_polyexp_t<32>(a, frames, s);
}

// ===================================================================== A3-A6 flow iteration
// flow_out = solve( window( M( R0, R1 sampled at p + flow_in, flow_in ) ) )
// One block = tw x th output pixels.  Phase 1 evaluates FarnebackUpdateMatrices for the
// tile plus its window halo (clamped coordinates = the replicate border of the blur)
// into LDS; phase 2/3 are the separable window; the 2x2 solve is in double like upstream.
__device__ __forceinline__ float2 rc_flow_in(const RcIterArgs& a, const float2* fin, int gx, int gy) {
    if (a.in_mode == 0) return make_float2(0.f, 0.f);
    if (a.in_mode == 1) return fin[(size_t)gy * a.w + gx];
    // resize(prevFlow, INTER_LINEAR) then flow *= 1/pyr_scale (optflow.cpp calc())
    float ax, ay;
    int sx = rc_src_x(gx, a.up_scale_x, a.fin_w, ax);
    int sx1 = min(sx + 1, a.fin_w - 1);
    int sy = rc_src_y(gy, a.up_scale_y, ay);
    int sy0 = rc_clampi(sy, 0, a.fin_h - 1), sy1 = rc_clampi(sy + 1, 0, a.fin_h - 1);
    const float2* S0 = fin + (size_t)sy0 * a.fin_w;
    const float2* S1 = fin + (size_t)sy1 * a.fin_w;
    float2 p00 = S0[sx], p01 = S0[sx1], p10 = S1[sx], p11 = S1[sx1];
    float a0 = 1.f - ax, a1 = ax, b0 = 1.f - ay, b1 = ay;
    float r0x = p00.x * a0 + p01.x * a1, r1x = p10.x * a0 + p11.x * a1;
    float r0y = p00.y * a0 + p01.y * a1, r1y = p10.y * a0 + p11.y * a1;
    float2 v;
    v.x = (r0x * b0 + r1x * b1) * a.up_mul;
    v.y = (r0y * b0 + r1y * b1) * a.up_mul;
    return v;
}

__global__ __launch_bounds__(RC_BLOCK) void k_flow_iter(RcIterArgs a) {
    extern __shared__ __align__(16) float smf[];
    const int tid = threadIdx.x;
    const int z = blockIdx.y;
    const int t = rc_xcd_remap(blockIdx.x, a.tiles_x * a.tiles_y);
    const int tx0 = (t % a.tiles_x) * a.tw, ty0 = (t / a.tiles_x) * a.th;
    const int w = a.w, h = a.h, m = a.win.m;
    const int MW = a.tw + 2 * m, MH = a.th + 2 * m, MP = MW | 1;
    float* Ms = smf;                    // [5][MH][MP]
    float* Vs = smf + 5 * MH * MP;      // [5][th][MP]

    const size_t s0 = (size_t)((a.slot0 + z) % a.nslots) * a.R_slot_stride;
    const size_t s1 = (size_t)((a.slot1 + z) % a.nslots) * a.R_slot_stride;
    const float4* RA0 = a.RA + s0;  const float* RB0 = a.RB + s0;
    const float4* RA1 = a.RA + s1;  const float* RB1 = a.RB + s1;
    const float2* fin = a.fin ? a.fin + (size_t)z * a.fin_pair_stride : nullptr;
    char* fout = a.fout + (size_t)z * a.fout_pair_stride;

    if (!a.solve) {
        for (int idx = tid; idx < a.tw * a.th; idx += RC_BLOCK) {
            int gx = tx0 + idx % a.tw, gy = ty0 + idx / a.tw;
            if (gx < w && gy < h)
                *(float2*)(fout + (size_t)gy * a.fout_step + (size_t)gx * 8) = rc_flow_in(a, fin, gx, gy);
        }
        return;
    }

    // ---- phase 1: matrices for tile + halo
    for (int idx = tid; idx < MW * MH; idx += RC_BLOCK) {
        int ly = idx / MW, lx = idx - ly * MW;
        int gx = rc_clampi(tx0 - m + lx, 0, w - 1), gy = rc_clampi(ty0 - m + ly, 0, h - 1);
        float2 d = rc_flow_in(a, fin, gx, gy);
        float dx = d.x, dy = d.y;
        float fx = gx + dx, fy = gy + dy;
        int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
        fx -= x1;
        fy -= y1;
        size_t p0 = (size_t)gy * w + gx;
        float4 A0 = RA0[p0];
        float B0 = RB0[p0];
        float r2, r3, r4, r5, r6;
        if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
            float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy);
            float a10 = (1.f - fx) * fy, a11 = fx * fy;
            size_t p = (size_t)y1 * w + x1;
            float4 q00 = RA1[p], q01 = RA1[p + 1], q10 = RA1[p + w], q11 = RA1[p + w + 1];
            float e00 = RB1[p], e01 = RB1[p + 1], e10 = RB1[p + w], e11 = RB1[p + w + 1];
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
        if ((unsigned)(gx - 5) >= (unsigned)(w - 10) || (unsigned)(gy - 5) >= (unsigned)(h - 10)) {
            // border[5] = {0.14, 0.14, 0.4472, 0.4472, 0.4472}
            float bl = gx < 5 ? (gx < 2 ? 0.14f : 0.4472f) : 1.f;
            int rx = w - gx - 1;
            float br = gx >= w - 5 ? (rx < 2 ? 0.14f : 0.4472f) : 1.f;
            float bt = gy < 5 ? (gy < 2 ? 0.14f : 0.4472f) : 1.f;
            int ry = h - gy - 1;
            float bb = gy >= h - 5 ? (ry < 2 ? 0.14f : 0.4472f) : 1.f;
            float scale = bl * br * bt * bb;
            r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
        }
        float* mp = Ms + ly * MP + lx;
        mp[0] = r4 * r4 + r6 * r6;
        mp[MH * MP] = (r4 + r5) * r6;
        mp[2 * MH * MP] = r5 * r5 + r6 * r6;
        mp[3 * MH * MP] = r4 * r2 + r6 * r3;
        mp[4 * MH * MP] = r6 * r2 + r5 * r3;
    }
    __syncthreads();

    // ---- phase 2: vertical window
    const int th = a.th, tw = a.tw;
    for (int idx = tid; idx < 5 * th * MW; idx += RC_BLOCK) {
        int c = idx / (th * MW), rem = idx - c * (th * MW);
        int o = rem / MW, col = rem - o * MW;
        const float* mc = Ms + c * MH * MP + (o + m) * MP + col;
        float s;
        if (a.win.gaussian) {
            s = mc[0] * a.win.k[0];
            for (int i = 1; i <= m; i++) s += (mc[i * MP] + mc[-i * MP]) * a.win.k[i];
        } else {
            s = mc[0];
            for (int i = 1; i <= m; i++) s += mc[i * MP] + mc[-i * MP];
        }
        Vs[c * th * MP + o * MP + col] = s;
    }
    __syncthreads();

    // ---- phase 3: horizontal window + solve
    for (int idx = tid; idx < tw * th; idx += RC_BLOCK) {
        int o = idx / tw, lx = idx - o * tw;
        int gx = tx0 + lx, gy = ty0 + o;
        if (gx >= w || gy >= h) continue;
        double g[5];
        const float* vc = Vs + o * MP + lx + m;
        if (a.win.gaussian) {
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const float* v = vc + c * th * MP;
                float s = v[0] * a.win.k[0];
                for (int i = 1; i <= m; i++) s += a.win.k[i] * (v[-i] + v[i]);
                g[c] = s;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const float* v = vc + c * th * MP;
                double s = v[0];
                for (int i = 1; i <= m; i++) s += (double)v[i] + (double)v[-i];
                g[c] = s * a.win.box_scale;
            }
        }
        double idet = 1. / (g[0] * g[2] - g[1] * g[1] + 1e-3);
        float2 f;
        f.x = (float)((g[0] * g[4] - g[1] * g[3]) * idet);
        f.y = (float)((g[2] * g[3] - g[1] * g[4]) * idet);
        *(float2*)(fout + (size_t)gy * a.fout_step + (size_t)gx * 8) = f;
    }
}

size_t rc_flow_iter_lds(int tw, int th, int m) {
    int MW = tw + 2 * m, MH = th + 2 * m, MP = MW | 1;
    return sizeof(float) * (5 * (size_t)MH * MP + 5 * (size_t)th * MP);
}

void rc_launch_flow_iter(const RcIterArgs& a, int pairs, hipStream_t s) {
    size_t lds = rc_flow_iter_lds(a.tw, a.th, a.win.m);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute((const void*)k_flow_iter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    dim3 grid(a.tiles_x * a.tiles_y, pairs, 1);
    hipLaunchKernelGGL(k_flow_iter, grid, dim3(RC_BLOCK), lds, s, a);
}

// ===================================================================== test helpers
__global__ void k_pack_R5(const float* R5, float4* RA, float* RB, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        RA[i] = make_float4(R5[5 * i], R5[5 * i + 1], R5[5 * i + 2], R5[5 * i + 3]);
        RB[i] = R5[5 * i + 4];
    }
}
__global__ void k_unpack_R5(const float4* RA, const float* RB, float* R5, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float4 v = RA[i];
        R5[5 * i] = v.x; R5[5 * i + 1] = v.y; R5[5 * i + 2] = v.z; R5[5 * i + 3] = v.w;
        R5[5 * i + 4] = RB[i];
    }
}
void rc_launch_pack_R5(const float* R5, float4* RA, float* RB, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_R5, dim3((n + 255) / 256), dim3(256), 0, s, R5, RA, RB, n);
}
void rc_launch_unpack_R5(const float4* RA, const float* RB, float* R5, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_unpack_R5, dim3((n + 255) / 256), dim3(256), 0, s, RA, RB, R5, n);
}
