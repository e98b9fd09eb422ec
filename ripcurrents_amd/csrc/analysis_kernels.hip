// analysis_kernels.hip -- gfx950 kernels and C ABI for the per-pixel analysis that consumes
// the flow field.  Reference arithmetic:
//   B1 ripcurrents.cpp:305-309 (cartToPolar, degrees)      B2 ripcurrents_module.cpp:89-144
//   B3 ripcurrents_module.cpp:153-212                       B4 ripcurrents_module.cpp:608-648
//   B5 ripcurrents_module.cpp:486-606,650-679; ripcurrents.cpp:656-698; pathlines.cpp:9-46
//   B7 ripcurrents_module.cpp:279-308,810-1015; main.cpp:1142-1153
//   B8 ripcurrents_module.cpp:1017-1138
// These are HBM-bound integer/float per-pixel passes: the polar conversion is fused into
// its consumers (never materialised), counts are privatised in LDS, thresholds stay on the
// device so the frame loop never synchronises with the host.
// Compiled with -ffp-contract=off: every float expression rounds as the reference's does,
// which makes bins, classes and particle positions bit-exact for identical flow input.

#include <cfloat>
#include <climits>
#include <cstring>
#include <vector>

#include "rc_host.h"

#define RC_BLOCK 256
#define RC_MAX_ADVECT_ITERATIONS 65536   // a kernel that never returns is a hung GPU, not a slow call
#define THR_UPPER 0
#define THR_UPPER2D 1
#define THR_PROP (1 + RC_HIST_DIRECTIONS)
#define THR_WORDS (1 + 2 * RC_HIST_DIRECTIONS)

// OpenCV core fastAtan32f (degrees), mathfuncs_core: 7th-order odd polynomial.
__device__ __forceinline__ float rc_fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y);
    // upstream branches on ax >= ay; both branches divide the smaller by the larger and run the same
    // polynomial, so one division and one polynomial on (min, max) give the same bits without the
    // compiler evaluating both sides
    const bool xmajor = ax >= ay;
    const float lo = xmajor ? ay : ax, hi = xmajor ? ax : ay;
    const float c = lo / (hi + (float)DBL_EPSILON);
    const float c2 = c * c;
    float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    if (!xmajor) a = 90.f - a;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// int angle = (a * HIST_DIRECTIONS) / 360 (ripcurrents_module.cpp:100); 36 (a == 360.0f, an
// out-of-bounds index in the reference) folds to direction 0.
__device__ __forceinline__ int rc_dir_index(float angle) {
    int d = rc_cvt_i32_x86((angle * RC_HIST_DIRECTIONS) / 360);
    if (d >= RC_HIST_DIRECTIONS || d < 0) d = 0;
    return d;
}

__device__ __forceinline__ const float2* rc_row2(const float* base, size_t step, int y) {
    return (const float2*)((const char*)base + (size_t)y * step);
}

// ============================================================================ B1+B2 histogram
// Block-private hist2d[36][50] in LDS; hist, histsum and histsum2d are its marginals and
// are formed when the block flushes.  Flow fields are smooth, so most lanes of a wave
// fall into a handful of bins: lanes are grouped by key with ballots and each group
// costs one LDS atomic (added with the group's population count).
#define RC_HIST_COPIES 16
template <int ROUNDS>
__device__ __forceinline__ void rc_hist_add(int* lh, int key) {
    // Peel off the most popular key(s) of the wave with scalar control flow (ballot masks in
    // SGPRs, the leader's key by v_readlane): a uniform field would otherwise serialise 64
    // same-address LDS atomics.  What is left takes plain per-lane LDS atomics (distinct keys
    // do not conflict); on textured flows more than one round cost more than it saved.
    unsigned long long todo = __ballot(key >= 0);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int round = 0; round < ROUNDS; round++) {
        if (!todo) break;
        int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
        int k = __builtin_amdgcn_readlane(key, leader);
        unsigned long long same = __ballot(key == k);
        if (lane == leader) atomicAdd(&lh[k], __popcll(same));
        if (key == k) key = -1;
        todo &= ~same;
    }
    if (key >= 0) atomicAdd(&lh[key], 1);
}

__device__ __forceinline__ int rc_hist_key(float2 f) {
    float mag = sqrtf(f.x * f.x + f.y * f.y);
    int bin = rc_cvt_i32_x86(mag * RC_HIST_RESOLUTION);      // NaN: INT_MIN, not counted
    if (!(bin < RC_HIST_BINS && bin >= 0)) return -1;
    return rc_dir_index(rc_fast_atan2_deg(f.y, f.x)) * RC_HIST_BINS + bin;
}


template <int ROUNDS>
__global__ __launch_bounds__(RC_BLOCK) void k_polar_hist(const float* flow0, size_t frame_stride, size_t step,
                                                         int w, int h, int* parts) {
    __shared__ int lh[RC_HIST_DIRECTIONS * RC_HIST_BINS];
    const float* flow = (const float*)((const char*)flow0 + (size_t)blockIdx.y * frame_stride);
    for (int i = threadIdx.x; i < RC_HIST_DIRECTIONS * RC_HIST_BINS; i += RC_BLOCK) lh[i] = 0;
    __syncthreads();
    const int w2 = (w + 1) >> 1;
    const int total = w2 * h;                 // the entry point bounds w * h
    const int span = (int)gridDim.x * RC_BLOCK;
    // every lane runs the same number of rounds so the ballots see whole waves; four loads are in
    // flight per thread before the first is consumed
    const int rounds = (total + span - 1) / span;
    // (row, pair-in-row) of the thread's item, advanced by `span` items per round without a division
    const int i0 = (int)blockIdx.x * RC_BLOCK + threadIdx.x;
    int yy = i0 / w2, xx = i0 - yy * w2;
    const int dy = span / w2, dx = span - dy * w2;
    constexpr int UNR = 4;
    // The next batch's loads are issued before the current batch is binned, so that a wave never computes
    // with nothing in flight (-4 %; the loads alone take 84 us per 32 1080p fields, the kernel 145: the
    // rest is instruction issue -- correctly rounded sqrt and divisions, ballots -- not memory)
    float4 vn[UNR];
    int nn[UNR];
    auto load_batch = [&](int it0) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            nn[u] = 0;
            vn[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int y = yy, x = xx * 2;
            yy += dy; xx += dx;
            if (xx >= w2) { xx -= w2; yy++; }
            if (it0 + u < rounds && y < h) {
                const float2* r = rc_row2(flow, step, y) + x;
                if (x + 1 < w && (((size_t)r) & 15) == 0) {
                    typedef float rc_f4 __attribute__((ext_vector_type(4)));
                    const rc_f4 t = __builtin_nontemporal_load((const rc_f4*)r);
                    vn[u] = make_float4(t.x, t.y, t.z, t.w);
                    nn[u] = 2;
                } else {
                    float2 a0 = r[0];
                    float2 a1 = x + 1 < w ? r[1] : make_float2(0.f, 0.f);
                    vn[u] = make_float4(a0.x, a0.y, a1.x, a1.y);
                    nn[u] = x + 1 < w ? 2 : 1;
                }
            }
        }
    };
    load_batch(0);
    for (int it0 = 0; it0 < rounds; it0 += UNR) {
        float4 v[UNR];
        int nv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) { v[u] = vn[u]; nv[u] = nn[u]; }
        if (it0 + UNR < rounds) load_batch(it0 + UNR);
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (it0 + u < rounds) {      // block-uniform
                int k0 = nv[u] >= 1 ? rc_hist_key(make_float2(v[u].x, v[u].y)) : -1;
                int k1 = nv[u] >= 2 ? rc_hist_key(make_float2(v[u].z, v[u].w)) : -1;
                rc_hist_add<ROUNDS>(lh, k0);
                rc_hist_add<ROUNDS>(lh, k1);
            }
        }
    }
    __syncthreads();
    // flush into one of RC_HIST_COPIES partial tables: blocks that share a table queue on
    // the same addresses, so the copies keep those chains short
    int* part = parts + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) % RC_HIST_COPIES) *
                            (RC_HIST_DIRECTIONS * RC_HIST_BINS);
    for (int i = threadIdx.x; i < RC_HIST_DIRECTIONS * RC_HIST_BINS; i += RC_BLOCK) {
        int v = lh[i];
        if (v) atomicAdd(&part[i], v);
    }
}

// ---- second form of the histogram kernel (the one that runs; k_polar_hist above is kept behind option
// "ablate" bit RC_ABL_HIST_V1 for A/B runs).  Two changes, same counts:
//  * the key.  ripcurrents.cpp:305-309 + ripcurrents_module.cpp:97-100 cost a correctly rounded sqrt and two IEEE
//    divisions per pixel (min/max inside fastAtan2, and (angle*36)/360).  Here the magnitude is s*rsq(s) and the
//    quotients are products with v_rcp_f32 / a constant -- each within a few ulps of the exact value -- and a pixel
//    whose scaled magnitude mag*20 or scaled angle lands within 1e-4 of an integer (where a few ulps could change
//    the floor), or is not finite, takes the exact path.  1e-4 is >= 5x the worst-case gap between the two
//    evaluations (relative 4e-7 on values below 51 and 36.0002), so the floors agree for every other pixel; about 4
//    pixels in 10 000 take the exact path.  tests: test_histogram_key_on_bin_edges, test_fast_atan_and_histogram_exact.
//  * the aggregation.  A lane takes a 2-pixel-wide column of 4 rows: 8 pixels that mostly share one key.  The lane
//    counts how many of its 8 keys equal its first one, adds the rest (rare) singly, and only the (key, count) pair
//    goes through the wave-level step: the first lane's key is broadcast, the counts of all lanes holding it are
//    summed with DPP / permute steps and added by one LDS atomic; lanes holding another key add their own pair.
//    One scalar round per 8 pixels instead of one per pixel.
#ifndef RC_HIST_ABL
#define RC_HIST_ABL 0     // diagnostic builds only (scripts/r3/variant_ana.sh)
#endif
__device__ __forceinline__ int rc_hist_key_fast(float2 f, bool& exact_needed) {
    const float s = f.x * f.x + f.y * f.y;
    const float m = __builtin_amdgcn_sqrtf(s);                       // v_sqrt_f32 (1 ulp); 0 -> 0, NaN -> NaN, inf -> inf
    const float t = m * (float)RC_HIST_RESOLUTION;
    // direction: OpenCV's polynomial on lo / hi with the quotient as a product (fused multiply-adds: this value only has
    // to land within a few ulps of the reference's, the guard below covers the rest)
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(f.x), ay = fabsf(f.y);
    const bool xmajor = ax >= ay;
    const float lo = xmajor ? ay : ax, hi = xmajor ? ax : ay;        // (selects: fminf / fmaxf add a canonicalising v_max each)
    const float c = lo * __builtin_amdgcn_rcpf(hi + (float)DBL_EPSILON);
    const float c2 = c * c;
    float a = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(p7, c2, p5), c2, p3), c2, p1) * c;
    if (!xmajor) a = 90.f - a;
    if (f.x < 0) a = 180.f - a;
    if (f.y < 0) a = 360.f - a;
    const float u = a * ((float)RC_HIST_DIRECTIONS / 360.f);
    // "within 1e-4 of an integer" on the fractional parts: |fract - 1/2| > 1/2 - 1e-4 (a NaN or an infinity fails the
    // comparisons below and takes the exact path too)
    const float gt = fabsf(__builtin_amdgcn_fractf(t) - 0.5f), gu = fabsf(__builtin_amdgcn_fractf(u) - 0.5f);
    const bool mag_ok = gt < 0.4999f;
    const bool dir_ok = gu < 0.4999f;
    const int bin = (int)t;                                          // v_cvt_i32_f32 saturates; t >= 0 or NaN (-> 0, flagged)
    const bool counted = bin < RC_HIST_BINS;
    exact_needed = !mag_ok || (counted && !dir_ok);
    // (int)u < 36 wherever dir_ok holds: u = 36 needs a >= 359.99999, whose fractional part fails the guard
    const int d = (int)u;
    return counted ? (int)__umul24((unsigned)d, (unsigned)RC_HIST_BINS) + bin : -1;
}

__device__ __forceinline__ int rc_wave_sum(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// PLAIN (decided by the launcher, rc_hist_rows_plain): every item whole and every row 16-byte aligned.
template <bool PLAIN>
__global__ __launch_bounds__(RC_BLOCK) void k_polar_hist_rows(const float* flow0, size_t frame_stride, size_t step,
                                                              int w, int h, int* parts) {
    __shared__ int lh[RC_HIST_DIRECTIONS * RC_HIST_BINS];
    const float* flow = (const float*)((const char*)flow0 + (size_t)blockIdx.y * frame_stride);
    for (int i = threadIdx.x; i < RC_HIST_DIRECTIONS * RC_HIST_BINS; i += RC_BLOCK) lh[i] = 0;
    __syncthreads();
    constexpr int NR = 4;                       // rows per item
    const int w2 = (w + 1) >> 1, hg = (h + NR - 1) / NR;
    const int total = w2 * hg;
    const int span = (int)gridDim.x * RC_BLOCK;
    const int rounds = (total + span - 1) / span;
    const int i0 = (int)blockIdx.x * RC_BLOCK + threadIdx.x;
    int gg = i0 / w2, xx = i0 - gg * w2;        // (row group, pair in row), advanced by span per round
    const int dg = span / w2, dx = span - dg * w2;
    const int lane = threadIdx.x & 63;
    // PLAIN: one 32-bit offset per item and a scalar base per row -- four loads and a handful of address instructions
    // where the general form spends ~120 on bounds, alignment and 64-bit row addresses.  A template parameter, not a
    // run-time flag: with both forms in one loop the number of loads in flight at the keys depends on the path taken
    // and the compiler waits for all of them (vmcnt(0)), i.e. for the prefetch it has just issued.
    constexpr bool plain = PLAIN;
    auto load_item = [&](int it, float4 (&vn)[NR], int (&nn)[NR]) {
        const int g = gg, x = xx * 2;
        gg += dg; xx += dx;
        if (xx >= w2) { xx -= w2; gg++; }
        if constexpr (plain) {
            // The four loads are issued by every lane in every round (a lane without an item reads the frame's first
            // texels): were they under a branch, the number of loads in flight at the keys below would depend on the
            // path and the compiler would have to wait for all of them -- the prefetch would overlap nothing.
            const bool valid = it < rounds && g < hg;
            const unsigned off = valid ? (unsigned)(g * NR) * (unsigned)step + (unsigned)x * 8u : 0u;
            typedef float rc_f4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int r = 0; r < NR; r++) {
                nn[r] = valid ? 2 : 0;
                const rc_f4 t = __builtin_nontemporal_load((const rc_f4*)((const char*)flow + (size_t)r * step + off));
                vn[r] = make_float4(t.x, t.y, t.z, t.w);
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            nn[r] = 0;
            vn[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int y = g * NR + r;
            if (it < rounds && g < hg && y < h) {
                const float2* p = rc_row2(flow, step, y) + x;
                if (x + 1 < w && (((size_t)p) & 15) == 0) {
                    typedef float rc_f4 __attribute__((ext_vector_type(4)));
                    const rc_f4 t = __builtin_nontemporal_load((const rc_f4*)p);
                    vn[r] = make_float4(t.x, t.y, t.z, t.w);
                    nn[r] = 2;
                } else {
                    const float2 a0 = p[0];
                    const float2 a1 = x + 1 < w ? p[1] : make_float2(0.f, 0.f);
                    vn[r] = make_float4(a0.x, a0.y, a1.x, a1.y);
                    nn[r] = x + 1 < w ? 2 : 1;
                }
            }
        }
    };
    auto process = [&](const float4 (&v)[NR], const int (&nv)[NR]) {
        int k[2 * NR];
        bool redo = false;
#if RC_HIST_ABL == 2     // timing-only build: the loads alone
        {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < NR; r++) acc += v[r].x + v[r].y + v[r].z + v[r].w;
            if (acc == 12345.678f) atomicAdd(&lh[0], 1);
            return;
        }
#endif
        unsigned em = 0;                         // bit i: key i has to be taken again with the reference's own arithmetic
        if constexpr (plain) {
            // (a lane without an item skips the keys altogether: whole waves do in the last round)
#pragma unroll
            for (int i = 0; i < 2 * NR; i++) k[i] = -1;
            if (nv[0]) {
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    bool e0, e1;
                    k[2 * r] = rc_hist_key_fast(make_float2(v[r].x, v[r].y), e0);
                    k[2 * r + 1] = rc_hist_key_fast(make_float2(v[r].z, v[r].w), e1);
                    em |= (e0 ? 1u : 0u) << (2 * r) | (e1 ? 1u : 0u) << (2 * r + 1);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                bool e0, e1;
                const int k0 = rc_hist_key_fast(make_float2(v[r].x, v[r].y), e0);
                const int k1 = rc_hist_key_fast(make_float2(v[r].z, v[r].w), e1);
                k[2 * r] = nv[r] >= 1 ? k0 : -1;
                k[2 * r + 1] = nv[r] >= 2 ? k1 : -1;
                em |= ((nv[r] >= 1 && e0) ? 1u : 0u) << (2 * r) | ((nv[r] >= 2 && e1) ? 1u : 0u) << (2 * r + 1);
            }
        }
        redo = em != 0;
        if (redo) {
            // a pixel near a bin / direction edge (about 4 in 10 000, i.e. one in every fifth wave-item): the reference's
            // own arithmetic, for the flagged pixels alone -- a position no lane has flagged is skipped by the whole wave
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (em >> (2 * r) & 1u) k[2 * r] = rc_hist_key(make_float2(v[r].x, v[r].y));
                if (em >> (2 * r + 1) & 1u) k[2 * r + 1] = rc_hist_key(make_float2(v[r].z, v[r].w));
            }
        }
#if RC_HIST_ABL == 1     // timing-only build: loads and keys, no aggregation
        {
            int acc = 0;
#pragma unroll
            for (int i = 0; i < 2 * NR; i++) acc ^= k[i];
            if (acc == 0x12345678) atomicAdd(&lh[0], 1);
            return;
        }
#endif
        // lane level: the first key and its multiplicity; other keys singly
        const int A = k[0];
        int cnt = 1;
#pragma unroll
        for (int i = 1; i < 2 * NR; i++) {
            if (k[i] == A) cnt++;
            else if (k[i] >= 0) atomicAdd(&lh[k[i]], 1);
        }
        // wave level: everything that shares the first active lane's key goes in one atomic
        const unsigned long long todo = __ballot(A >= 0);
        if (todo) {
            const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
            const int kL = __builtin_amdgcn_readlane(A, leader);
            const bool same = A == kL;
            const int sum = rc_wave_sum(same ? cnt : 0);
            if (lane == leader) atomicAdd(&lh[kL], sum);
            else if (A >= 0 && !same) atomicAdd(&lh[A], cnt);
        }
    };
    // Two register sets, filled in turn one item ahead (no copy at the loop's end: a copy of registers that loads are
    // still writing would have to wait for those loads).  Block-uniform trip count: the wave-level steps see whole waves.
    float4 va[NR], vb[NR];
    int na[NR], nb[NR];
    load_item(0, va, na);
    for (int it = 0; it < rounds; it += 2) {
        if (plain || it + 1 < rounds) load_item(it + 1, vb, nb);     // (plain: past the last round the loads are dummies)
        process(va, na);
        if (it + 1 < rounds) {
            if (plain || it + 2 < rounds) load_item(it + 2, va, na);
            process(vb, nb);
        }
    }
    __syncthreads();
    int* part = parts + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) % RC_HIST_COPIES) *
                            (RC_HIST_DIRECTIONS * RC_HIST_BINS);
    for (int i = threadIdx.x; i < RC_HIST_DIRECTIONS * RC_HIST_BINS; i += RC_BLOCK) {
        int v = lh[i];
        if (v) atomicAdd(&part[i], v);
    }
}

// Folds (and clears) the partial tables into the slot's cumulative counters
// words = hist[50] | hist2d[1800] | histsum | histsum2d[36]; hist, histsum and histsum2d
// are the marginals of hist2d (ripcurrents_module.cpp:102-104 increments all four together).
__global__ __launch_bounds__(RC_BLOCK) void k_hist_fold(int* parts, int* words) {
    __shared__ int lbin[RC_HIST_BINS], ldir[RC_HIST_DIRECTIONS], ltot;
    if (threadIdx.x < RC_HIST_BINS) lbin[threadIdx.x] = 0;
    if (threadIdx.x < RC_HIST_DIRECTIONS) ldir[threadIdx.x] = 0;
    if (threadIdx.x == 0) ltot = 0;
    __syncthreads();
    const int i = blockIdx.x * RC_BLOCK + threadIdx.x;
    int v = 0;
    if (i < RC_HIST_DIRECTIONS * RC_HIST_BINS) {
        int t[RC_HIST_COPIES];
#pragma unroll
        for (int c = 0; c < RC_HIST_COPIES; c++) t[c] = parts[c * (RC_HIST_DIRECTIONS * RC_HIST_BINS) + i];
#pragma unroll
        for (int c = 0; c < RC_HIST_COPIES; c++) {
            v += t[c];
            if (t[c]) parts[c * (RC_HIST_DIRECTIONS * RC_HIST_BINS) + i] = 0;
        }
        if (v) {
            words[RC_HIST_BINS + i] += v;
            atomicAdd(&lbin[i % RC_HIST_BINS], v);
            atomicAdd(&ldir[i / RC_HIST_BINS], v);
            atomicAdd(&ltot, v);
        }
    }
    __syncthreads();
    int* hist = words;
    int* histsum = words + RC_HIST_BINS + RC_HIST_DIRECTIONS * RC_HIST_BINS;
    int* histsum2d = histsum + 1;
    if (threadIdx.x < RC_HIST_BINS && lbin[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lbin[threadIdx.x]);
    if (threadIdx.x >= 64 && threadIdx.x < 64 + RC_HIST_DIRECTIONS && ldir[threadIdx.x - 64])
        atomicAdd(&histsum2d[threadIdx.x - 64], ldir[threadIdx.x - 64]);
    if (threadIdx.x == 128 && ltot) atomicAdd(histsum, ltot);
}

// Threshold scans, ripcurrents_module.cpp:109-144, one lane per direction.
__global__ __launch_bounds__(RC_BLOCK) void k_thresholds(const int* words, float* thr) {
    // the 1887 counters are staged in LDS in one parallel sweep: the scans below are chains of
    // dependent reads (a global round trip each would cost ~10 us for the kernel)
    __shared__ int sw[RC_HIST_WORDS];
    for (int i = threadIdx.x; i < RC_HIST_WORDS; i += RC_BLOCK) sw[i] = words[i];
    __syncthreads();
    const int* hist = sw;
    const int* hist2d = sw + RC_HIST_BINS;
    const int histsum = sw[RC_HIST_BINS + RC_HIST_DIRECTIONS * RC_HIST_BINS];
    const int* histsum2d = sw + RC_HIST_BINS + RC_HIST_DIRECTIONS * RC_HIST_BINS + 1;
    __shared__ int s_target, s_threshsum;
    if (threadIdx.x == 0) {
        int threshsum = 0, bin = RC_HIST_BINS - 1;
        while (threshsum < (histsum * .05)) {
            threshsum += hist[bin];
            bin--;
        }
        thr[THR_UPPER] = bin / float(RC_HIST_RESOLUTION);
        s_target = bin;
        s_threshsum = threshsum;
    }
    __syncthreads();
    int angle = threadIdx.x;
    if (angle < RC_HIST_DIRECTIONS) {
        int threshsum2 = 0, bin = RC_HIST_BINS - 1;
        while (threshsum2 < (histsum2d[angle] * .05)) {
            threshsum2 += hist2d[angle * RC_HIST_BINS + bin];
            bin--;
        }
        float u = bin / float(RC_HIST_RESOLUTION);
        if (u < 0.01) u = 0.01;
        thr[THR_UPPER2D + angle] = u;
        int threshsum3 = 0;
        bin = RC_HIST_BINS - 1;
        while (bin > s_target) {
            threshsum3 += hist2d[angle * RC_HIST_BINS + bin];
            bin--;
        }
        thr[THR_PROP + angle] = ((float)threshsum3) / s_threshsum;
    }
}

// ============================================================================ B1+B3 classify + accumulate
struct ClassifyArgs {
    const float* flow; size_t flow_step;
    int w, h, framecount;
    const int* d_framecount;      // non-null: the frame counter lives on the device (rcflow_frame_loop_step)
    float MID, LOWER;
    const float* thr;
    float* acc;
    float* polar; size_t polar_step;
    float* wclass; size_t wc_step;
    float* out; size_t out_step;
    uint8_t* mask; size_t mask_step;
};

__global__ __launch_bounds__(RC_BLOCK) void k_classify_accumulate(ClassifyArgs a) {
    __shared__ float s_u2d[RC_HIST_DIRECTIONS];
    if (threadIdx.x < RC_HIST_DIRECTIONS) s_u2d[threadIdx.x] = a.thr[THR_UPPER2D + threadIdx.x];
    __syncthreads();
    const float UPPER = a.thr[THR_UPPER];
    const int framecount = a.d_framecount ? *a.d_framecount : a.framecount;
    const long long total = (long long)a.w * a.h;
    for (long long i = (long long)blockIdx.x * RC_BLOCK + threadIdx.x; i < total;
         i += (long long)gridDim.x * RC_BLOCK) {
        int y = (int)(i / a.w), x = (int)(i - (long long)y * a.w);
        float2 f = rc_row2(a.flow, a.flow_step, y)[x];
        float val = sqrtf(f.x * f.x + f.y * f.y);
        float ang = rc_fast_atan2_deg(f.y, f.x);
        int dir = rc_dir_index(ang);
        // create_flow: ripcurrents_module.cpp:162-168
        float cx = 0.f, cy = 0.f, cz = 0.f, inc = 0.f;
        if (val > UPPER) { cx = .5f; inc = 1.f; }
        else if (val > a.MID) { cz = 1.f; }
        else if (val > a.LOWER) { cz = .5f; }
        else { cy = .5f; }
        if (a.wclass) {
            float* p = (float*)((char*)a.wclass + (size_t)y * a.wc_step) + 3 * x;
            p[0] = cx; p[1] = cy; p[2] = cz;
        }
        if (a.polar) {
            float zz = val / s_u2d[dir];
            float* p = (float*)((char*)a.polar + (size_t)y * a.polar_step) + 3 * x;
            p[0] = ang; p[1] = zz > 1 ? 1.f : .7f; p[2] = zz;
        }
        // create_accumulationbuffer: ripcurrents_module.cpp:191-211
        float acc = a.acc[i];
        if (framecount > 30) {
            acc = inc + acc;
            a.acc[i] = acc;
        }
        int v = (int)acc;
        float ox = 0.f, oy = 0.f, oz = 0.f;
        uint8_t mk = 0;
        if (v > .1 * framecount) {
            if (v < .2 * framecount) oz = 1.f;
            else ox = 1.f;
        } else {
            oy = .5f;
            mk = 255;
        }
        if (a.out) {
            float* p = (float*)((char*)a.out + (size_t)y * a.out_step) + 3 * x;
            p[0] = ox; p[1] = oy; p[2] = oz;
        }
        if (a.mask) a.mask[(size_t)y * a.mask_step + x] = mk;
    }
}

// ============================================================================ B4/B5 advection
// Bilinear sampler shared by every streamline variant (ripcurrents_module.cpp:494-508).
__device__ __forceinline__ bool rc_sample_flow(const float* flow, size_t step, int w, int h, float x, float y,
                                               float& dx, float& dy) {
    int xind = rc_cvt_i32_x86(floorf(x)), yind = rc_cvt_i32_x86(floorf(y));
    float xrem = x - xind, yrem = y - yind;
    if (xind < 1 || yind < 1 || xind + 2 > w || yind + 2 > h) return false;
    const float2* r0 = rc_row2(flow, step, yind) + xind;
    const float2* r1 = rc_row2(flow, step, yind + 1) + xind;
    float2 p00 = r0[0], p01 = r0[1], p10 = r1[0], p11 = r1[1];
    float wa = 1 - xrem, wb = 1 - yrem;
    dx = p00.x * wa * wb + p01.x * xrem * wb + p10.x * wa * yrem + p11.x * xrem * yrem;
    dy = p00.y * wa * wb + p01.y * xrem * wb + p10.y * wa * yrem + p11.y * xrem * yrem;
    return true;
}

__global__ __launch_bounds__(RC_BLOCK) void k_advect_field(float2* pt, float* dist, const float* flow,
                                                           size_t step, int w, int h, float dt, int iterations,
                                                           float UPPER_arg, const float* thr) {
    const float UPPER = UPPER_arg < 0 ? thr[THR_UPPER] : UPPER_arg;
    const long long total = (long long)w * h;
    for (long long i = (long long)blockIdx.x * RC_BLOCK + threadIdx.x; i < total;
         i += (long long)gridDim.x * RC_BLOCK) {
        int yo = (int)(i / w), xo = (int)(i - (long long)yo * w);
        float2 p = pt[i];
        float d = dist[i];
        for (int it = 0; it < iterations; it++) {
            float x = p.x + xo, y = p.y + yo;
            float dx, dy;
            if (!rc_sample_flow(flow, step, w, h, x, y, dx, dy)) break;
            float r = sqrtf(dx * dx + dy * dy);
            if (r > UPPER) break;
            p.x = p.x + dx * dt / iterations;
            p.y = p.y + dy * dt / iterations;
            d = d + r;
        }
        pt[i] = p;
        dist[i] = d;
    }
}

__global__ void k_advect_points(float2* pts, int n, const float* flow, size_t step, int w, int h, float dt,
                                int iterations, float UPPER_arg, const float* thr, int variant, float2* trace) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float UPPER = UPPER_arg < 0 ? thr[THR_UPPER] : UPPER_arg;
    const int iters = variant == 2 ? 100 : iterations;
    float2 p = pts[s];
    bool alive = true;
    for (int i = 0; i < iters; i++) {
        float dx, dy;
        if (alive && !rc_sample_flow(flow, step, w, h, p.x, p.y, dx, dy)) alive = false;
        if (alive) {
            float r = sqrtf(dx * dx + dy * dy);
            if ((variant == 0 || variant == 3) && r > UPPER) alive = false;
            if (variant == 1 && r > 5) alive = false;
        }
        if (alive) {
            if (variant <= 1) { p.x = p.x + dx * dt; p.y = p.y + dy * dt; }
            else if (variant == 2) { p.x = p.x + (float)(dx * 0.1); p.y = p.y + (float)(dy * 0.1); }
            else { p.x = p.x + dx * dt / iterations; p.y = p.y + dy * dt / iterations; }
        }
        if (trace) trace[(size_t)s * iters + i] = p;
    }
    pts[s] = p;
}

__global__ __launch_bounds__(RC_BLOCK) void k_get_delta_field(float* pt, size_t pt_step, const float* flow,
                                                              size_t step, int w, int h, float dt, float UPPER_arg,
                                                              const float* thr) {
    const float UPPER = UPPER_arg < 0 ? thr[THR_UPPER] : UPPER_arg;
    const long long total = (long long)w * h;
    for (long long i = (long long)blockIdx.x * RC_BLOCK + threadIdx.x; i < total;
         i += (long long)gridDim.x * RC_BLOCK) {
        int yo = (int)(i / w), xo = (int)(i - (long long)yo * w);
        float2* pp = (float2*)((char*)pt + (size_t)yo * pt_step) + xo;
        float2 p = *pp;
        float dx, dy;
        if (!rc_sample_flow(flow, step, w, h, p.x + xo, p.y + yo, dx, dy)) continue;
        float r = sqrtf(dx * dx + dy * dy);
        if (r > UPPER) continue;
        p.x = p.x + dx * dt;
        p.y = p.y + dy * dt;
        *pp = p;
    }
}

// ============================================================================ B7 post-ops
// Two-stage deterministic reductions: per-block partials in double, then one block folds
// them in a fixed order.  part[b] = (sum x, sum y, sum |f|, max |f|)
__device__ __forceinline__ double rc_wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float rc_wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}

__global__ __launch_bounds__(RC_BLOCK) void k_flow_stats_partial(const float* flow, size_t step, int x0, int y0,
                                                                 int w, int h, double* part) {
    double sx = 0, sy = 0, sm = 0;
    float mx = 0.f;
    const long long total = (long long)w * h;
    for (long long i = (long long)blockIdx.x * RC_BLOCK + threadIdx.x; i < total;
         i += (long long)gridDim.x * RC_BLOCK) {
        int y = (int)(i / w), x = (int)(i - (long long)y * w);
        float2 f = rc_row2(flow, step, y0 + y)[x0 + x];
        float mag = sqrtf(f.x * f.x + f.y * f.y);
        sx += f.x; sy += f.y; sm += mag;
        mx = fmaxf(mx, mag);
    }
    __shared__ double sh[4][4];
    sx = rc_wave_sum(sx); sy = rc_wave_sum(sy); sm = rc_wave_sum(sm); mx = rc_wave_max(mx);
    int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[wv][0] = sx; sh[wv][1] = sy; sh[wv][2] = sm; sh[wv][3] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0, c = 0, d = 0;
        for (int k = 0; k < 4; k++) { a += sh[k][0]; b += sh[k][1]; c += sh[k][2]; d = fmax(d, sh[k][3]); }
        part[4 * blockIdx.x] = a; part[4 * blockIdx.x + 1] = b; part[4 * blockIdx.x + 2] = c;
        part[4 * blockIdx.x + 3] = d;
    }
}
__global__ void k_flow_stats_final(const double* part, int nblocks, double* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a = 0, b = 0, c = 0, d = 0;
        for (int k = 0; k < nblocks; k++) {
            a += part[4 * k]; b += part[4 * k + 1]; c += part[4 * k + 2]; d = fmax(d, part[4 * k + 3]);
        }
        out[0] = a; out[1] = b; out[2] = c; out[3] = d;
    }
}

// mode 0: subtructAverage (:810-898); 1: subtructMeanMagnitude (:900-1015); 2: stabilizer (:300-307)
__global__ __launch_bounds__(RC_BLOCK) void k_flow_postop(float* flow, size_t step, int w, int h, int mode,
                                                          const double* stats, double divx, double divy) {
    const long long total = (long long)w * h;
    const double a0 = stats[0] / divx, a1 = stats[1] / divy;
    const float meanval = (float)(stats[2] / ((double)w * h));
    for (long long i = (long long)blockIdx.x * RC_BLOCK + threadIdx.x; i < total;
         i += (long long)gridDim.x * RC_BLOCK) {
        int y = (int)(i / w), x = (int)(i - (long long)y * w);
        float2* pp = (float2*)((char*)flow + (size_t)y * step) + x;
        float2 f = *pp;
        if (mode == 0) {
            f.x = (float)(f.x - a0);
            f.y = (float)(f.y - a1);
        } else if (mode == 1) {
            float mag = sqrtf(f.x * f.x + f.y * f.y);
            float ux, uy;
            if (mag == 0) { ux = 0.f; uy = 0.f; }
            else { ux = f.x / mag; uy = f.y / mag; }
            f.x = ux * (mag - meanval);
            f.y = uy * (mag - meanval);
        } else {
            if (f.x != 0) f.x = (float)(f.x - a0 * 0.2);
            if (f.y != 0) f.y = (float)(f.y - a1 * 0.2);
        }
        *pp = f;
    }
}

// main.cpp:1142-1153: avg -= slot/window; slot = cur; avg += slot/window
__global__ void k_window_mean(float* avg, float* slot, const float* cur, size_t n, float inv) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float t = slot[i] * inv;
        float a = avg[i] - t;
        float c = cur[i];
        slot[i] = c;
        t = c * inv;
        avg[i] = a + t;
    }
}

// ============================================================================ B8 colouring
// `uchar = float` as x86 compiles it: cvttss2si then the low byte (NaN/overflow -> INT_MIN).
__device__ __forceinline__ uint8_t rc_f2u8(float v) {
    int iv;
    if (!(v > -2147483904.f && v < 2147483648.f)) iv = INT_MIN;
    else iv = (int)v;
    return (uint8_t)(iv & 0xFF);
}

__global__ __launch_bounds__(RC_BLOCK) void k_vector_to_color(const float* flow, size_t step, int w, int h,
                                                              uint8_t* hsv, size_t hsv_step, float max_disp) {
    const long long total = (long long)w * h;
    for (long long i = (long long)blockIdx.x * RC_BLOCK + threadIdx.x; i < total;
         i += (long long)gridDim.x * RC_BLOCK) {
        int y = (int)(i / w), x = (int)(i - (long long)y * w);
        float2 f = rc_row2(flow, step, y)[x];
        // ripcurrents_module.cpp:1030 calls libm's float atan2, whose last bit differs from platform to platform (and from
        // the device's atan2f): the angle is evaluated in double and rounded once -- the correctly rounded float, which is
        // what the oracle pins too -- so the truncation to the hue byte below cannot flip
        float theta = (float)((float)atan2((double)f.y, (double)f.x) * 180 / 3.14159265358979323846);
        theta += theta < 0 ? 360 : 0;
        float mag = sqrtf(f.x * f.x + f.y * f.y);
        uint8_t* q = hsv + (size_t)y * hsv_step + 3 * x;
        q[0] = rc_f2u8(theta / 2);
        q[1] = 255;
        q[2] = rc_f2u8(mag * 255 / max_disp);
    }
}

__global__ __launch_bounds__(RC_BLOCK) void k_shear_to_color(const float* flow, size_t step, int w, int h,
                                                             uint8_t* hsv, size_t hsv_step, float max_fro,
                                                             float* part_max) {
    const int off = 10;
    const int iw = w - 2 * off, ih = h - 2 * off;
    float mx = 0.f;
    const long long total = (long long)iw * ih;
    for (long long i = (long long)blockIdx.x * RC_BLOCK + threadIdx.x; i < total;
         i += (long long)gridDim.x * RC_BLOCK) {
        int row = off + (int)(i / iw), col = off + (int)(i % iw);
        float2 above = rc_row2(flow, step, row - off)[col], below = rc_row2(flow, step, row + off)[col];
        float2 left = rc_row2(flow, step, row)[col - off], right = rc_row2(flow, step, row)[col + off];
        float j00 = right.x - left.x, j01 = above.x - below.x;
        float j10 = right.y - left.y, j11 = above.y - below.y;
        float fro = j00 * j00 + j01 * j01 + j10 * j10 + j11 * j11;
        fro = sqrtf(fro);
        uint8_t* q = hsv + (size_t)row * hsv_step + 3 * col;
        q[0] = rc_f2u8(128 - fro * 128 / max_fro);
        q[1] = 255;
        q[2] = 255;
        mx = fmaxf(mx, fro);
    }
    __shared__ float sh[4];
    mx = rc_wave_max(mx);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) part_max[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// ============================================================================ section 8(f) next rows
// create_edges (ripcurrents_module.cpp:216-220, ripcurrents.cpp:477-479): dilate with the
// 5x5 MORPH_ELLIPSE element, then the morphological gradient (dilate - erode) of that result,
// fused: a 64x16 tile loads the mask with a 4-pixel halo, forms the first dilation on
// tile+2 in LDS, then both second-stage operators.  Pixels outside the image never win
// (OpenCV's default constant border: minimum for dilate, maximum for erode).
__device__ __forceinline__ bool rc_ellipse5(int i, int j) {   // getStructuringElement(MORPH_ELLIPSE, 5x5)
    return (i == 0 || i == 4) ? (j == 2) : true;
}

__global__ __launch_bounds__(RC_BLOCK) void k_create_edges(const uint8_t* mask, size_t mask_step, int w, int h,
                                                           uint8_t* out, size_t out_step) {
    constexpr int TW = 64, TH = 16, AW = TW + 8, AH = TH + 8, BW = TW + 4, BH = TH + 4;
    __shared__ uint8_t A[AH][AW];      // mask, tile + 4
    __shared__ uint8_t B[BH][BW];      // first dilation, tile + 2
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    for (int idx = threadIdx.x; idx < AW * AH; idx += RC_BLOCK) {
        int ly = idx / AW, lx = idx - ly * AW;
        int gx = tx0 - 4 + lx, gy = ty0 - 4 + ly;
        A[ly][lx] = ((unsigned)gx < (unsigned)w && (unsigned)gy < (unsigned)h) ? mask[(size_t)gy * mask_step + gx] : 0;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < BW * BH; idx += RC_BLOCK) {
        int ly = idx / BW, lx = idx - ly * BW;
        int v = 0;
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 5; j++)
                if (rc_ellipse5(i, j)) v = max(v, (int)A[ly + i][lx + j]);
        B[ly][lx] = (uint8_t)v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TW * TH; idx += RC_BLOCK) {
        int ly = idx / TW, lx = idx - ly * TW;
        int gx = tx0 + lx, gy = ty0 + ly;
        if (gx >= w || gy >= h) continue;
        int d = 0, e = 255;
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 5; j++)
                if (rc_ellipse5(i, j)) {
                    int yy = gy + i - 2, xx = gx + j - 2;
                    if ((unsigned)xx < (unsigned)w && (unsigned)yy < (unsigned)h) {
                        int v = B[ly + i][lx + j];
                        d = max(d, v);
                        e = min(e, v);
                    }
                }
        out[(size_t)gy * out_step + gx] = (uint8_t)(d - e);
    }
}

// Frame pre-processing (ripcurrents.cpp:209-210): resize(8UC3, INTER_LINEAR) with resize.cpp's
// 11-bit fixed-point coefficients, then cvtColor(BGR2GRAY) with its 14-bit ones; one thread
// per output pixel, integer arithmetic throughout (bit-exact).
__global__ __launch_bounds__(RC_BLOCK) void k_resize_bgr_to_gray(const uint8_t* bgr, size_t step, int sw, int sh,
                                                                 uint8_t* gray, size_t gray_step, int dw, int dh,
                                                                 double scale_x, double scale_y) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
    const int sx1 = min(sx + 1, sw - 1);
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
    const int a0 = __float2int_rn((1.f - fx) * 2048.f), a1 = __float2int_rn(fx * 2048.f);
    const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
    const uint8_t* S0 = bgr + (size_t)sy0 * step;
    const uint8_t* S1 = bgr + (size_t)sy1 * step;
    int px[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int h0 = S0[sx * 3 + c] * a0 + S0[sx1 * 3 + c] * a1;
        int h1 = S1[sx * 3 + c] * a0 + S1[sx1 * 3 + c] * a1;
        int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        px[c] = min(max(v, 0), 255);
    }
    gray[(size_t)dy * gray_step + dx] = (uint8_t)((px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + (1 << 13)) >> 14);
}

// ============================================================================ host side
static int grid_for(long long items) {
    long long b = (items + RC_BLOCK - 1) / RC_BLOCK;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));   // memory-bound: cap and grid-stride
}

static int analysis_ensure(rc_ctx* ctx, RcSlot& s, int w, int h, bool reset) {
    if (w <= 0 || h <= 0) return RC_EINVAL;
    if (w > ctx->max_w || h > ctx->max_h) { rc_set_error("frame exceeds the context size"); return RC_ESIZE; }
    RcAnalysis& an = s.an;
    bool fresh = (an.w != w || an.h != h || !an.hist.p);
    if (!fresh && !reset) return RC_OK;
    size_t n = (size_t)w * h;
    int rc;
    if ((rc = rc_buf_ensure(an.hist, RC_HIST_WORDS * sizeof(int)))) return rc;
    if ((rc = rc_buf_ensure(an.hist_part, RC_HIST_COPIES * RC_HIST_DIRECTIONS * RC_HIST_BINS * sizeof(int)))) return rc;
    if ((rc = rc_buf_ensure(an.thr, (THR_WORDS + 3) * sizeof(float)))) return rc;
    if ((rc = rc_buf_ensure(an.acc, n * sizeof(float)))) return rc;
    if ((rc = rc_buf_ensure(an.pt, n * sizeof(float2)))) return rc;
    if ((rc = rc_buf_ensure(an.dist, n * sizeof(float)))) return rc;
    if ((rc = rc_buf_ensure(an.scratch, (4 * 2048 + 8) * sizeof(double)))) return rc;
    an.w = w; an.h = h;
    an.hist_added = 0;
    RC_HIP(hipMemsetAsync(an.hist.p, 0, RC_HIST_WORDS * sizeof(int), s.cur));
    RC_HIP(hipMemsetAsync(an.hist_part.p, 0, RC_HIST_COPIES * RC_HIST_DIRECTIONS * RC_HIST_BINS * sizeof(int), s.cur));
    RC_HIP(hipMemsetAsync(an.acc.p, 0, n * sizeof(float), s.cur));
    RC_HIP(hipMemsetAsync(an.pt.p, 0, n * sizeof(float2), s.cur));
    RC_HIP(hipMemsetAsync(an.dist.p, 0, n * sizeof(float), s.cur));
    // UPPER = 100.0, UPPER2d = prop_above_upper = 0 (ripcurrents.cpp:149-154)
    float init[THR_WORDS] = {0};
    init[THR_UPPER] = 100.0f;
    RC_HIP(hipMemcpyAsync(an.thr.p, init, sizeof(init), hipMemcpyHostToDevice, s.cur));
    RC_HIP(hipStreamSynchronize(s.cur));
    return RC_OK;
}

extern "C" int rcflow_analysis_reset(rc_ctx* ctx, int stream, int w, int h) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    RC_HIP(hipSetDevice(ctx->device));
    return analysis_ensure(ctx, *s, w, h, true);
}

static int check_flow(const float* d_flow, size_t step, int w, int h) {
    if (!d_flow || w <= 0 || h <= 0 || step < (size_t)w * 8 || (step & 7)) {
        rc_set_error("bad flow field argument");
        return RC_EINVAL;
    }
    return RC_OK;
}

extern "C" int rcflow_histogram_clip_dev(rc_ctx* ctx, int stream, const float* d_flows, size_t flow_frame_stride,
                                         size_t flow_step, int count, int w, int h) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    int rc = check_flow(d_flows, flow_step, w, h);
    if (rc) return rc;
    if (count < 1 || count > 65535 || (count > 1 && flow_frame_stride < flow_step * (size_t)(h - 1) + (size_t)w * 8))
        return RC_EINVAL;
    if ((long long)w * h > 0x3fffffffll) { rc_set_error("frame too large for the histogram kernel's 32-bit item index"); return RC_ESIZE; }
    RC_HIP(hipSetDevice(ctx->device));
    if ((rc = analysis_ensure(ctx, *s, w, h, false))) return rc;
    // The counters are the reference's `int` (ripcurrents.cpp:147-150, never reset there: they wrap after
    // 2^31 / (w h) frames, about four minutes of 640x480 video).  Every pixel adds at most one count, so the
    // pixels added since the last reset bound histsum: refuse the call that could wrap it.
    if (s->an.hist_added + (long long)w * h * count > 0x7fffffffll) {
        rc_set_error("flow histogram would exceed its int32 counters (%lld pixels counted since the last reset): "
                     "start a new segment with rcflow_histogram_reset_dev", s->an.hist_added);
        return RC_ESTATE;
    }
    s->an.hist_added += (long long)w * h * count;
    {
        RcProfScope ps(ctx, s->cur, RC_K_HIST, 0, 8. * w * h * count);
        long long per_frame = ((long long)(w + 1) / 2) * h;
        int nb = grid_for(per_frame);
        // many light blocks: the partial tables keep the flush chains short, and one ballot round catches the dominant
        // bin of a wave.  8192 blocks per launch = four items per thread at 32 frames of 1080p: one more round of prefetch
        // than 16384 (134 -> 129 us), 4096: 131, 32768: 147 (profiles/r03_notes.md)
        int cap = (ctx->hist_blocks > 0 ? ctx->hist_blocks : 8192) / count;
        if (cap < 8) cap = 8;
        if (nb > cap) nb = cap;
        if (ctx->ablate & RC_ABL_HIST_V1) {
            hipLaunchKernelGGL(k_polar_hist<1>, dim3(nb, count), dim3(RC_BLOCK), 0, s->cur, d_flows, flow_frame_stride,
                               flow_step, w, h, (int*)s->an.hist_part.p);
        } else {
            // items are 2 x 4 pixel columns: a quarter of the items of the first form for the same block count
            int nb4 = grid_for(((long long)(w + 1) / 2) * ((h + 3) / 4));
            if (nb4 > cap) nb4 = cap;
            // every item whole (even width, height a multiple of the item's four rows), every row of every frame 16-byte
            // aligned, offsets inside a frame below 4 GB: the kernel's plain form
            const bool plain = ((w & 1) | (h & 3)) == 0 && ((((size_t)d_flows) | flow_step | (count > 1 ? flow_frame_stride : 0)) & 15) == 0 &&
                               (size_t)h * flow_step < ((size_t)1 << 32);
            if (plain)
                hipLaunchKernelGGL(k_polar_hist_rows<true>, dim3(nb4, count), dim3(RC_BLOCK), 0, s->cur, d_flows, flow_frame_stride,
                                   flow_step, w, h, (int*)s->an.hist_part.p);
            else
                hipLaunchKernelGGL(k_polar_hist_rows<false>, dim3(nb4, count), dim3(RC_BLOCK), 0, s->cur, d_flows, flow_frame_stride,
                                   flow_step, w, h, (int*)s->an.hist_part.p);
        }
        hipLaunchKernelGGL(k_hist_fold, dim3((RC_HIST_DIRECTIONS * RC_HIST_BINS + RC_BLOCK - 1) / RC_BLOCK),
                           dim3(RC_BLOCK), 0, s->cur, (int*)s->an.hist_part.p, (int*)s->an.hist.p);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_histogram_dev(rc_ctx* ctx, int stream, const float* d_flow, size_t flow_step, int w, int h) {
    return rcflow_histogram_clip_dev(ctx, stream, d_flow, 0, flow_step, 1, w, h);
}

extern "C" int rcflow_thresholds_dev(rc_ctx* ctx, int stream) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!s->an.hist.p) { rc_set_error("no histogram state: call rcflow_analysis_reset first"); return RC_ESTATE; }
    RC_HIP(hipSetDevice(ctx->device));
    {
        RcProfScope ps(ctx, s->cur, RC_K_THRESH, 0, 4. * RC_HIST_WORDS);
        hipLaunchKernelGGL(k_thresholds, dim3(1), dim3(RC_BLOCK), 0, s->cur, (const int*)s->an.hist.p, (float*)s->an.thr.p);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

// Thresholds from a caller-held block of counters (the all-reduced global histogram of SURVEY 8(e)):
// the slot's own cumulative counters stay local, UPPER / UPPER2d / prop_above_upper become global.
extern "C" int rcflow_thresholds_words_dev(rc_ctx* ctx, int stream, const int32_t* d_words) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_words) return RC_EINVAL;
    if (!s->an.thr.p) { rc_set_error("no histogram state: call rcflow_analysis_reset first"); return RC_ESTATE; }
    RC_HIP(hipSetDevice(ctx->device));
    {
        RcProfScope ps(ctx, s->cur, RC_K_THRESH, 0, 4. * RC_HIST_WORDS);
        hipLaunchKernelGGL(k_thresholds, dim3(1), dim3(RC_BLOCK), 0, s->cur, (const int*)d_words, (float*)s->an.thr.p);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_histogram_read(rc_ctx* ctx, int stream, int32_t* hist, int32_t* hist2d, int32_t* histsum,
                                     int32_t* histsum2d, float* UPPER, float* UPPER2d, float* prop) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!s->an.hist.p) { rc_set_error("no histogram state"); return RC_ESTATE; }
    RC_HIP(hipSetDevice(ctx->device));
    int32_t words[RC_HIST_WORDS];
    float thr[THR_WORDS];
    RC_HIP(hipMemcpyAsync(words, s->an.hist.p, sizeof(words), hipMemcpyDeviceToHost, s->cur));
    RC_HIP(hipMemcpyAsync(thr, s->an.thr.p, sizeof(thr), hipMemcpyDeviceToHost, s->cur));
    RC_HIP(hipStreamSynchronize(s->cur));
    if (hist) memcpy(hist, words, RC_HIST_BINS * 4);
    if (hist2d) memcpy(hist2d, words + RC_HIST_BINS, RC_HIST_DIRECTIONS * RC_HIST_BINS * 4);
    if (histsum) *histsum = words[RC_HIST_BINS + RC_HIST_DIRECTIONS * RC_HIST_BINS];
    if (histsum2d) memcpy(histsum2d, words + RC_HIST_BINS + RC_HIST_DIRECTIONS * RC_HIST_BINS + 1, RC_HIST_DIRECTIONS * 4);
    if (UPPER) *UPPER = thr[THR_UPPER];
    if (UPPER2d) memcpy(UPPER2d, thr + THR_UPPER2D, RC_HIST_DIRECTIONS * 4);
    if (prop) memcpy(prop, thr + THR_PROP, RC_HIST_DIRECTIONS * 4);
    return RC_OK;
}

extern "C" int rcflow_histogram_write(rc_ctx* ctx, int stream, const int32_t* words) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !words) return RC_EINVAL;
    if (!s->an.hist.p) { rc_set_error("no histogram state"); return RC_ESTATE; }
    RC_HIP(hipSetDevice(ctx->device));
    RC_HIP(hipMemcpyAsync(s->an.hist.p, words, RC_HIST_WORDS * 4, hipMemcpyHostToDevice, s->cur));
    RC_HIP(hipStreamSynchronize(s->cur));
    s->an.hist_added = words[RC_HIST_BINS + RC_HIST_DIRECTIONS * RC_HIST_BINS];     // histsum
    return RC_OK;
}

// Starts a new segment: zeroes the cumulative counters on the slot's stream (thresholds, accumulator and
// particles are left alone).  Asynchronous.
extern "C" int rcflow_histogram_reset_dev(rc_ctx* ctx, int stream) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!s->an.hist.p) { rc_set_error("no histogram state: call rcflow_analysis_reset first"); return RC_ESTATE; }
    RC_HIP(hipSetDevice(ctx->device));
    RC_HIP(hipMemsetAsync(s->an.hist.p, 0, RC_HIST_WORDS * sizeof(int), s->cur));
    s->an.hist_added = 0;
    return RC_OK;
}

extern "C" int rcflow_histogram_device_ptr(rc_ctx* ctx, int stream, int32_t** d_words) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_words) return RC_EINVAL;
    if (!s->an.hist.p) { rc_set_error("no histogram state"); return RC_ESTATE; }
    *d_words = (int32_t*)s->an.hist.p;
    return RC_OK;
}

int rc_analysis_ensure(rc_ctx* ctx, RcSlot& s, int w, int h) { return analysis_ensure(ctx, s, w, h, false); }

// framecount < 0: the counter is the slot's device word (an.loopc), incremented by k_loop_count in the same stream
int rc_classify_accumulate(rc_ctx* ctx, int stream, const float* d_flow, size_t flow_step, int w, int h, int framecount,
                           float MID, float LOWER, float* d_polar, size_t polar_step, float* d_wclass, size_t wc_step,
                           float* d_out, size_t out_step, uint8_t* d_mask, size_t mask_step) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    int rc = check_flow(d_flow, flow_step, w, h);
    if (rc) return rc;
    if ((d_polar && polar_step < (size_t)w * 12) || (d_wclass && wc_step < (size_t)w * 12) ||
        (d_out && out_step < (size_t)w * 12) || (d_mask && mask_step < (size_t)w)) {
        rc_set_error("output step smaller than a row");
        return RC_EINVAL;
    }
    RC_HIP(hipSetDevice(ctx->device));
    if ((rc = analysis_ensure(ctx, *s, w, h, false))) return rc;
    const int* d_fc = framecount < 0 ? (const int*)s->an.loopc.p : nullptr;
    if (framecount < 0 && !d_fc) { rc_set_error("no device frame counter"); return RC_ESTATE; }
    ClassifyArgs a = {d_flow, flow_step, w, h, framecount, d_fc, MID, LOWER, (const float*)s->an.thr.p,
                      (float*)s->an.acc.p, d_polar, polar_step, d_wclass, wc_step, d_out, out_step, d_mask, mask_step};
    {
        RcProfScope ps(ctx, s->cur, RC_K_CLASSIFY, 0, 16. * w * h);
        hipLaunchKernelGGL(k_classify_accumulate, dim3(grid_for((long long)w * h)), dim3(RC_BLOCK), 0, s->cur, a);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_classify_accumulate_dev(rc_ctx* ctx, int stream, const float* d_flow, size_t flow_step,
                                              int w, int h, int framecount, float MID, float LOWER,
                                              float* d_polar, size_t polar_step, float* d_wclass, size_t wc_step,
                                              float* d_out, size_t out_step, uint8_t* d_mask, size_t mask_step) {
    if (framecount < 0) { rc_set_error("negative framecount"); return RC_EINVAL; }
    return rc_classify_accumulate(ctx, stream, d_flow, flow_step, w, h, framecount, MID, LOWER, d_polar, polar_step, d_wclass,
                                  wc_step, d_out, out_step, d_mask, mask_step);
}

// the frame counter of rcflow_frame_loop_step: set (host value) or advanced by one, in stream order
__global__ void k_loop_count(int* c, int set, int value) { *c = set ? value : *c + 1; }
int rc_loop_counter(rc_ctx* ctx, RcSlot& s, bool set, int value) {
    int rc;
    if ((rc = rc_buf_ensure(s.an.loopc, 16))) return rc;
    hipLaunchKernelGGL(k_loop_count, dim3(1), dim3(1), 0, s.cur, (int*)s.an.loopc.p, set ? 1 : 0, value);
    RC_HIP(hipGetLastError());
    (void)ctx;
    return RC_OK;
}
// what rcflow_histogram_dev books on the host for one more w x h field (the graph replay of rcflow_frame_loop_step
// runs the kernels without passing through it)
int rc_hist_book(RcSlot& s, int w, int h, bool commit) {
    if (s.an.hist_added + (long long)w * h > 0x7fffffffll) {
        rc_set_error("the flow histogram's int32 counters would wrap (%lld pixels counted): "
                     "start a new segment with rcflow_histogram_reset_dev", s.an.hist_added);
        return RC_ESTATE;
    }
    if (commit) s.an.hist_added += (long long)w * h;
    return RC_OK;
}

extern "C" int rcflow_accumulator_read(rc_ctx* ctx, int stream, float* acc) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !acc) return RC_EINVAL;
    if (!s->an.acc.p) { rc_set_error("no analysis state"); return RC_ESTATE; }
    RC_HIP(hipSetDevice(ctx->device));
    RC_HIP(hipMemcpyAsync(acc, s->an.acc.p, (size_t)s->an.w * s->an.h * 4, hipMemcpyDeviceToHost, s->cur));
    RC_HIP(hipStreamSynchronize(s->cur));
    return RC_OK;
}

extern "C" int rcflow_advect_field_dev(rc_ctx* ctx, int stream, const float* d_flow, size_t flow_step, int w, int h,
                                       float dt, int iterations, float UPPER) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    int rc = check_flow(d_flow, flow_step, w, h);
    if (rc) return rc;
    if (iterations < 0 || iterations > RC_MAX_ADVECT_ITERATIONS) return RC_EINVAL;   // the reference passes 1 or 100
    RC_HIP(hipSetDevice(ctx->device));
    if ((rc = analysis_ensure(ctx, *s, w, h, false))) return rc;
    {
        RcProfScope ps(ctx, s->cur, RC_K_ADVECT_FIELD, 0, 32. * w * h);
        hipLaunchKernelGGL(k_advect_field, dim3(grid_for((long long)w * h)), dim3(RC_BLOCK), 0, s->cur,
                           (float2*)s->an.pt.p, (float*)s->an.dist.p, d_flow, flow_step, w, h, dt, iterations, UPPER,
                           (const float*)s->an.thr.p);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_advect_field_read(rc_ctx* ctx, int stream, float* pt_xy, float* dist) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!s->an.pt.p) { rc_set_error("no analysis state"); return RC_ESTATE; }
    RC_HIP(hipSetDevice(ctx->device));
    size_t n = (size_t)s->an.w * s->an.h;
    if (pt_xy) RC_HIP(hipMemcpyAsync(pt_xy, s->an.pt.p, n * 8, hipMemcpyDeviceToHost, s->cur));
    if (dist) RC_HIP(hipMemcpyAsync(dist, s->an.dist.p, n * 4, hipMemcpyDeviceToHost, s->cur));
    RC_HIP(hipStreamSynchronize(s->cur));
    return RC_OK;
}

static int thr_ptr(rc_ctx* ctx, RcSlot& s, float UPPER, const float** thr) {
    *thr = (const float*)s.an.thr.p;
    if (UPPER < 0 && !s.an.thr.p) { rc_set_error("UPPER<0 needs the slot's analysis state"); return RC_ESTATE; }
    (void)ctx;
    return RC_OK;
}

extern "C" int rcflow_advect_points_dev(rc_ctx* ctx, int stream, float* d_pts, int n, const float* d_flow,
                                        size_t flow_step, int w, int h, float dt, int iterations, float UPPER,
                                        int variant, float* d_trace) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    int rc = check_flow(d_flow, flow_step, w, h);
    if (rc) return rc;
    if (n < 0 || (n && !d_pts) || iterations < 0 || iterations > RC_MAX_ADVECT_ITERATIONS || variant < 0 || variant > 4)
        return RC_EINVAL;
    if (n == 0) return RC_OK;
    RC_HIP(hipSetDevice(ctx->device));
    const float* thr;
    if ((rc = thr_ptr(ctx, *s, UPPER, &thr))) return rc;
    {
        RcProfScope ps(ctx, s->cur, RC_K_ADVECT_POINTS, 0, 16. * n);
        hipLaunchKernelGGL(k_advect_points, dim3((n + 63) / 64), dim3(64), 0, s->cur, (float2*)d_pts, n, d_flow,
                           flow_step, w, h, dt, iterations, UPPER, thr, variant, (float2*)d_trace);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_get_delta_field_dev(rc_ctx* ctx, int stream, float* d_pt, size_t pt_step, const float* d_flow,
                                          size_t flow_step, int w, int h, float dt, float UPPER) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    int rc = check_flow(d_flow, flow_step, w, h);
    if (rc) return rc;
    if (!d_pt || pt_step < (size_t)w * 8) return RC_EINVAL;
    RC_HIP(hipSetDevice(ctx->device));
    const float* thr;
    if ((rc = thr_ptr(ctx, *s, UPPER, &thr))) return rc;
    hipLaunchKernelGGL(k_get_delta_field, dim3(grid_for((long long)w * h)), dim3(RC_BLOCK), 0, s->cur, d_pt, pt_step,
                       d_flow, flow_step, w, h, dt, UPPER, thr);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

// stats of a sub-rectangle into scratch[4*2048 .. +4): sum x, sum y, sum |f|, max |f|
static int flow_stats(rc_ctx* ctx, RcSlot& s, const float* d_flow, size_t step, int x0, int y0, int w, int h) {
    int rc = rc_buf_ensure(s.an.scratch, (4 * 2048 + 8) * sizeof(double));
    if (rc) return rc;
    double* part = (double*)s.an.scratch.p;
    int nb = grid_for((long long)w * h);
    hipLaunchKernelGGL(k_flow_stats_partial, dim3(nb), dim3(RC_BLOCK), 0, s.cur, d_flow, step, x0, y0, w, h, part);
    hipLaunchKernelGGL(k_flow_stats_final, dim3(1), dim3(64), 0, s.cur, part, nb, part + 4 * 2048);
    (void)ctx;
    return RC_OK;
}

static int postop(rc_ctx* ctx, int stream, float* d_flow, size_t step, int w, int h, int mode) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    int rc = check_flow(d_flow, step, w, h);
    if (rc) return rc;
    RC_HIP(hipSetDevice(ctx->device));
    double divx = (double)w * h, divy = (double)w * h;
    RcProfScope ps(ctx, s->cur, RC_K_POSTOP, mode, 24. * w * h);
    if (mode == 2) {
        // patch = last 10% of rows and columns; the reference divides sum_x by the patch's
        // column count and sum_y by its row count (ripcurrents_module.cpp:295-296)
        int x0 = (int)(w * 0.9), y0 = (int)(h * 0.9);
        divx = w - x0;
        divy = h - y0;
        if ((rc = flow_stats(ctx, *s, d_flow, step, x0, y0, w - x0, h - y0))) return rc;
    } else {
        if ((rc = flow_stats(ctx, *s, d_flow, step, 0, 0, w, h))) return rc;
    }
    hipLaunchKernelGGL(k_flow_postop, dim3(grid_for((long long)w * h)), dim3(RC_BLOCK), 0, s->cur, d_flow, step, w, h,
                       mode, (const double*)s->an.scratch.p + 4 * 2048, divx, divy);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_subtract_average_dev(rc_ctx* ctx, int stream, float* d_flow, size_t step, int w, int h) {
    return postop(ctx, stream, d_flow, step, w, h, 0);
}
extern "C" int rcflow_subtract_mean_magnitude_dev(rc_ctx* ctx, int stream, float* d_flow, size_t step, int w, int h) {
    return postop(ctx, stream, d_flow, step, w, h, 1);
}
extern "C" int rcflow_stabilizer_dev(rc_ctx* ctx, int stream, float* d_flow, size_t step, int w, int h) {
    return postop(ctx, stream, d_flow, step, w, h, 2);
}

extern "C" int rcflow_window_mean_dev(rc_ctx* ctx, int stream, float* d_avg, float* d_slot, const float* d_cur,
                                      size_t n, int window) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !d_avg || !d_slot || !d_cur || window < 1) return RC_EINVAL;
    RC_HIP(hipSetDevice(ctx->device));
    float inv = (float)(1. / (float)window);
    hipLaunchKernelGGL(k_window_mean, dim3(grid_for((long long)n)), dim3(RC_BLOCK), 0, s->cur, d_avg, d_slot, d_cur, n, inv);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_vector_to_color_dev(rc_ctx* ctx, int stream, const float* d_flow, size_t step, int w, int h,
                                          uint8_t* d_hsv, size_t hsv_step, float* max_io) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    int rc = check_flow(d_flow, step, w, h);
    if (rc) return rc;
    if (!d_hsv || hsv_step < (size_t)w * 3 || !max_io) return RC_EINVAL;
    RC_HIP(hipSetDevice(ctx->device));
    if ((rc = flow_stats(ctx, *s, d_flow, step, 0, 0, w, h))) return rc;
    {
        RcProfScope ps(ctx, s->cur, RC_K_COLOR, 0, 11. * w * h);
        hipLaunchKernelGGL(k_vector_to_color, dim3(grid_for((long long)w * h)), dim3(RC_BLOCK), 0, s->cur, d_flow, step,
                           w, h, d_hsv, hsv_step, *max_io);
    }
    double st[4];
    RC_HIP(hipMemcpyAsync(st, (const double*)s->an.scratch.p + 4 * 2048, sizeof(st), hipMemcpyDeviceToHost, s->cur));
    RC_HIP(hipStreamSynchronize(s->cur));
    *max_io = (float)st[3];
    return RC_OK;
}

extern "C" int rcflow_shear_rate_to_color_dev(rc_ctx* ctx, int stream, const float* d_flow, size_t step, int w, int h,
                                              uint8_t* d_hsv, size_t hsv_step, float* max_io) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    int rc = check_flow(d_flow, step, w, h);
    if (rc) return rc;
    if (!d_hsv || hsv_step < (size_t)w * 3 || !max_io) return RC_EINVAL;
    if (w <= 20 || h <= 20) { *max_io = 0.f; return RC_OK; }
    RC_HIP(hipSetDevice(ctx->device));
    if ((rc = rc_buf_ensure(s->an.scratch, (4 * 2048 + 8) * sizeof(double)))) return rc;
    int nb = grid_for((long long)(w - 20) * (h - 20));
    float* part = (float*)s->an.scratch.p;
    {
        RcProfScope ps(ctx, s->cur, RC_K_COLOR, 1, 11. * w * h);
        hipLaunchKernelGGL(k_shear_to_color, dim3(nb), dim3(RC_BLOCK), 0, s->cur, d_flow, step, w, h, d_hsv, hsv_step,
                           *max_io, part);
    }
    std::vector<float> hp(nb);
    RC_HIP(hipMemcpyAsync(hp.data(), part, nb * sizeof(float), hipMemcpyDeviceToHost, s->cur));
    RC_HIP(hipStreamSynchronize(s->cur));
    float mx = 0.f;
    for (float v : hp) mx = v > mx ? v : mx;
    *max_io = mx;
    return RC_OK;
}

// ---------------------------------------------------------------------------- section 8(f) entry points
extern "C" int rcflow_create_edges_dev(rc_ctx* ctx, int stream, const uint8_t* d_mask, size_t mask_step, int w, int h,
                                       uint8_t* d_out, size_t out_step) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!d_mask || !d_out || w <= 0 || h <= 0 || mask_step < (size_t)w || out_step < (size_t)w || d_mask == d_out) {
        rc_set_error("bad mask arguments (in-place is not supported)");
        return RC_EINVAL;
    }
    RC_HIP(hipSetDevice(ctx->device));
    {
        RcProfScope ps(ctx, s->cur, RC_K_EDGES, 0, 2. * w * h);
        hipLaunchKernelGGL(k_create_edges, dim3((w + 63) / 64, (h + 15) / 16), dim3(RC_BLOCK), 0, s->cur, d_mask, mask_step,
                           w, h, d_out, out_step);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_resize_bgr_to_gray_dev(rc_ctx* ctx, int stream, const uint8_t* d_bgr, size_t step, int sw, int sh,
                                             uint8_t* d_gray, size_t gray_step, int dw, int dh) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!d_bgr || !d_gray || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || step < (size_t)sw * 3 || gray_step < (size_t)dw) {
        rc_set_error("bad frame arguments");
        return RC_EINVAL;
    }
    RC_HIP(hipSetDevice(ctx->device));
    double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    {
        RcProfScope ps(ctx, s->cur, RC_K_PREPROC, 0, 3. * sw * sh + 1. * dw * dh);
        hipLaunchKernelGGL(k_resize_bgr_to_gray, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(RC_BLOCK), 0, s->cur, d_bgr, step,
                           sw, sh, d_gray, gray_step, dw, dh, scale_x, scale_y);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

// ===================================================================== SURVEY 8(f) row 4: display path
// ripcurrents.cpp:233-273 (ripcurrents_module.cpp:13-60) and :405.  Only 8-bit BGR images (and the
// float "flow" window) leave the device.
//   k_display_max     minMaxLoc maximum of |pt|, dist or |pt|/dist (NaNs never win)          -> scratch
//   k_display_map     convertTo(8U, 255/max) + applyColorMap(COLORMAP_JET)                    -> 8UC3
//   k_positions       streamline_positions scatter                                            -> 32FC3
//   k_hsv_to_bgr      cvtColor(32FC3, CV_HSV2BGR)                                             -> 32FC3
__device__ __forceinline__ float rc_display_value(const float2* pt, const float* dist, size_t i, int which) {
    const float2 p = pt[i];
    const float mag = sqrtf(p.x * p.x + p.y * p.y);          // cv::magnitude
    return which == 0 ? mag : (which == 1 ? dist[i] : mag / dist[i]);
}

__global__ __launch_bounds__(RC_BLOCK) void k_display_max(const float2* __restrict__ pt, const float* __restrict__ dist,
                                                          size_t n, int which, unsigned int* out) {
    __shared__ float red[RC_BLOCK];
    float m = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * RC_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * RC_BLOCK) {
        float v = rc_display_value(pt, dist, i, which);
        if (v > m) m = v;                                      // false for NaN, like minMaxLoc
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = RC_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
        __syncthreads();
    }
    // every value is >= 0 (or -inf when nothing compared greater): order-preserving integer key
    if (threadIdx.x == 0) {
        float r = red[0];
        unsigned int key = r >= 0.f ? __float_as_uint(r) + 0x80000000u : ~__float_as_uint(r);
        atomicMax(out, key);
    }
}

__global__ __launch_bounds__(RC_BLOCK) void k_display_map(const float2* __restrict__ pt, const float* __restrict__ dist,
                                                          int w, int h, int which, const unsigned int* maxkey,
                                                          const uint8_t* __restrict__ lut, uint8_t* bgr, size_t bgr_step) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const unsigned int key = *maxkey;
    const float mx = __uint_as_float(key & 0x80000000u ? key - 0x80000000u : ~key);
    const float alpha = (float)(255 / (double)mx);
    const float v = rc_display_value(pt, dist, (size_t)y * w + x, which) * alpha;
    // saturate_cast<uchar>(cvRound(v)): NaN / out-of-range convert to INT_MIN -> 0
    int idx = fabsf(v) < 2147483648.f ? (int)rintf(v) : (int)0x80000000;
    idx = idx < 0 ? 0 : (idx > 255 ? 255 : idx);
    uint8_t* o = bgr + (size_t)y * bgr_step + 3 * x;
    o[0] = lut[3 * idx]; o[1] = lut[3 * idx + 1]; o[2] = lut[3 * idx + 2];
}

__global__ __launch_bounds__(RC_BLOCK) void k_positions(const float2* __restrict__ pt, int w, int h, float* density,
                                                        size_t density_step) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float2 p = pt[(size_t)y * w + x];
    const int xind = rc_cvt_i32_x86(roundf(floorf(p.x + x))), yind = rc_cvt_i32_x86(roundf(floorf(p.y + y)));
    if (xind < 1 || yind < 1 || xind + 2 > w || yind + 2 > h) return;
    float* d = (float*)((char*)density + (size_t)yind * density_step) + 3 * xind;
    d[0] = 1.f; d[1] = 1.f; d[2] = 1.f;        // every writer stores the same value: order is irrelevant
}

__global__ __launch_bounds__(RC_BLOCK) void k_hsv_to_bgr(const float* __restrict__ hsv, size_t hsv_step, int w, int h,
                                                         float* bgr, size_t bgr_step) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float* s = (const float*)((const char*)hsv + (size_t)y * hsv_step) + 3 * x;
    float hh = s[0];
    const float ss = s[1], vv = s[2];
    float b, g, r;
    if (ss == 0) b = g = r = vv;
    else {
        hh *= 6.f / 360.f;
        // color_hsv.cpp wraps the hue with `do h -= 6; while (h >= 6)`, which never ends for an infinite or
        // huge hue (x - 6 == x).  Same steps here, at most 64 of them (|hue| < 3840 degrees behaves like
        // upstream); a hue still out of range after that is taken as 0 instead of hanging.
        for (int it = 0; it < 64 && hh < 0; it++) hh += 6;
        for (int it = 0; it < 64 && hh >= 6; it++) hh -= 6;
        if (hh < 0 || hh >= 6) hh = 0.f;
        int sector = rc_cvt_i32_x86(floorf(hh));
        hh -= sector;
        if ((unsigned)sector >= 6u) { sector = 0; hh = 0.f; }
        const float t0 = vv, t1 = vv * (1.f - ss), t2 = vv * (1.f - ss * hh), t3 = vv * (1.f - ss * (1.f - hh));
        // sector_data = {1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}  (b, g, r)
        switch (sector) {
            case 0: b = t1; g = t3; r = t0; break;
            case 1: b = t1; g = t0; r = t2; break;
            case 2: b = t3; g = t0; r = t1; break;
            case 3: b = t0; g = t2; r = t1; break;
            case 4: b = t0; g = t1; r = t3; break;
            default: b = t2; g = t1; r = t0; break;
        }
    }
    float* d = (float*)((char*)bgr + (size_t)y * bgr_step) + 3 * x;
    d[0] = b; d[1] = g; d[2] = r;
}

// applyColorMap(COLORMAP_JET): colormap.cpp tabulates clip(1.5 - |4x - c|, 0, 1), x = i/255, c = 3, 2, 1
// for r, g, b as float literals and passes them through linear_colormap (float interp1 onto
// linspace(0,1,256), then convertTo(8U, 255)).
extern "C" int rcflow_jet_lut(uint8_t* lut_bgr) {
    if (!lut_bgr) return RC_EINVAL;
    float tab[3][256], X[256];
    const float step = (1.f - 0.f) / (256 - 1);
    for (int i = 0; i < 256; i++) {
        double x = i / 255.0;
        X[i] = 0.f + i * step;
        for (int c = 0; c < 3; c++) {           // b, g, r
            double v = 1.5 - fabs(4 * x - (c + 1));
            tab[c][i] = (float)(v < 0 ? 0 : (v > 1 ? 1 : v));
        }
    }
    for (int i = 0; i < 256; i++) {
        const float xi = X[i];
        int low = 0, high = 255;
        if (xi < X[low]) high = 1;
        if (xi > X[high]) low = high - 1;
        while (high - low > 1) {
            int c = low + ((high - low) >> 1);
            if (xi > X[c]) low = c; else high = c;
        }
        for (int c = 0; c < 3; c++) {
            const float* Y = tab[c];
            float yi = Y[low] + (xi - X[low]) * (Y[high] - Y[low]) / (X[high] - X[low]);
            int q = (int)nearbyintf(yi * 255.f);
            lut_bgr[3 * i + c] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
    }
    return RC_OK;
}

static int display_lut(RcSlot& s) {
    if (s.an.jet.p) return RC_OK;
    int rc = rc_buf_ensure(s.an.jet, 768 + 16);
    if (rc) return rc;
    uint8_t lut[768];
    rcflow_jet_lut(lut);
    RC_HIP(hipMemcpy(s.an.jet.p, lut, 768, hipMemcpyHostToDevice));
    return RC_OK;
}

extern "C" int rcflow_streamline_display_dev(rc_ctx* ctx, int stream, int which, uint8_t* d_bgr, size_t bgr_step,
                                             float* max_out) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!s->an.pt.p || !s->an.dist.p) { rc_set_error("no streamline field in this slot (rcflow_advect_field_dev first)"); return RC_ESTATE; }
    const int w = s->an.w, h = s->an.h;
    if (which < 0 || which > 2 || !d_bgr || bgr_step < (size_t)w * 3) { rc_set_error("bad display arguments"); return RC_EINVAL; }
    RC_HIP(hipSetDevice(ctx->device));
    int rc = display_lut(*s);
    if (rc) return rc;
    unsigned int* key = (unsigned int*)((uint8_t*)s->an.jet.p + 768);
    RC_HIP(hipMemsetAsync(key, 0, 4, s->cur));
    const size_t n = (size_t)w * h;
    {
        RcProfScope ps(ctx, s->cur, RC_K_DISPLAY, 0, 27. * n);
        hipLaunchKernelGGL(k_display_max, dim3(1024), dim3(RC_BLOCK), 0, s->cur, (const float2*)s->an.pt.p,
                           (const float*)s->an.dist.p, n, which, key);
        hipLaunchKernelGGL(k_display_map, dim3((w + 63) / 64, (h + 3) / 4), dim3(RC_BLOCK), 0, s->cur,
                           (const float2*)s->an.pt.p, (const float*)s->an.dist.p, w, h, which, key,
                           (const uint8_t*)s->an.jet.p, d_bgr, bgr_step);
    }
    RC_HIP(hipGetLastError());
    if (max_out) {
        unsigned int k = 0;
        RC_HIP(hipMemcpyAsync(&k, key, 4, hipMemcpyDeviceToHost, s->cur));
        RC_HIP(hipStreamSynchronize(s->cur));
        unsigned int bits = (k & 0x80000000u) ? k - 0x80000000u : ~k;
        memcpy(max_out, &bits, 4);
    }
    return RC_OK;
}

extern "C" int rcflow_streamline_positions_dev(rc_ctx* ctx, int stream, float* d_density, size_t density_step) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!s->an.pt.p) { rc_set_error("no streamline field in this slot"); return RC_ESTATE; }
    const int w = s->an.w, h = s->an.h;
    if (!d_density || density_step < (size_t)w * 12) return RC_EINVAL;
    RC_HIP(hipSetDevice(ctx->device));
    {
        RcProfScope ps(ctx, s->cur, RC_K_DISPLAY, 0, 20. * w * h);
        hipLaunchKernelGGL(k_positions, dim3((w + 63) / 64, (h + 3) / 4), dim3(RC_BLOCK), 0, s->cur, (const float2*)s->an.pt.p,
                           w, h, d_density, density_step);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_hsv_to_bgr_dev(rc_ctx* ctx, int stream, const float* d_hsv, size_t hsv_step, int w, int h,
                                     float* d_bgr, size_t bgr_step) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!d_hsv || !d_bgr || w <= 0 || h <= 0 || hsv_step < (size_t)w * 12 || bgr_step < (size_t)w * 12) return RC_EINVAL;
    RC_HIP(hipSetDevice(ctx->device));
    {
        RcProfScope ps(ctx, s->cur, RC_K_HSV2BGR, 0, 24. * w * h);
        hipLaunchKernelGGL(k_hsv_to_bgr, dim3((w + 63) / 64, (h + 3) / 4), dim3(RC_BLOCK), 0, s->cur, d_hsv, hsv_step, w, h,
                           d_bgr, bgr_step);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

extern "C" int rcflow_analysis_size(rc_ctx* ctx, int stream, int* w, int* h) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s || !w || !h) return RC_EINVAL;
    *w = s->an.w; *h = s->an.h;
    return RC_OK;
}

// ===================================================================== INTER_AREA pre-processing
// resize(frame, Size(dw, dh), 0, 0, INTER_AREA) + cvtColor(BGR2GRAY) for the first frame of a run
// (ripcurrents.cpp:186, main.cpp:126, ...).  resize.cpp: integer factors -> resizeAreaFast_ (sum of the
// area; 2 x 2: (sum + 2) >> 2 as its SIMD path does, else saturate_cast(sum * (1.f / area))); otherwise
// resizeArea_ with the DecimateAlpha tables of computeResizeAreaTab: per source row a float row
// buffer buf += S * alpha (table order), then sum = beta * buf for the first row of an output row and
// sum += beta * buf after it.  One thread per output pixel reproduces exactly that order.
struct RcAreaArgs {
    const uint8_t* bgr; size_t step;
    int sw, sh, dw, dh;
    int fast, iscale_x, iscale_y;
    const int* xstart; const int* xsi; const float* xalpha;    // xstart has dw + 1 entries
    const int* ystart; const int* ysi; const float* yalpha;
    uint8_t* gray; size_t gray_step;
};

__global__ __launch_bounds__(RC_BLOCK) void k_resize_area_bgr_to_gray(RcAreaArgs a) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= a.dw || dy >= a.dh) return;
    int px[3];
    if (a.fast) {
        int sum[3] = {0, 0, 0};
        for (int ky = 0; ky < a.iscale_y; ky++) {
            const uint8_t* S = a.bgr + (size_t)(dy * a.iscale_y + ky) * a.step + (size_t)dx * a.iscale_x * 3;
            for (int kx = 0; kx < a.iscale_x; kx++) { sum[0] += S[3 * kx]; sum[1] += S[3 * kx + 1]; sum[2] += S[3 * kx + 2]; }
        }
        const float scale = 1.f / (float)(a.iscale_x * a.iscale_y);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int v = (a.iscale_x == 2 && a.iscale_y == 2) ? (sum[c] + 2) >> 2 : (int)rintf((float)sum[c] * scale);
            px[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
    } else {
        float sum[3] = {0.f, 0.f, 0.f};
        const int x0 = a.xstart[dx], x1 = a.xstart[dx + 1], y0 = a.ystart[dy], y1 = a.ystart[dy + 1];
        for (int j = y0; j < y1; j++) {
            const uint8_t* S = a.bgr + (size_t)a.ysi[j] * a.step;
            const float beta = a.yalpha[j];
            float buf[3] = {0.f, 0.f, 0.f};
            for (int k = x0; k < x1; k++) {
                const uint8_t* p = S + 3 * a.xsi[k];
                const float al = a.xalpha[k];
                buf[0] += (float)p[0] * al; buf[1] += (float)p[1] * al; buf[2] += (float)p[2] * al;
            }
            if (j == y0) { sum[0] = beta * buf[0]; sum[1] = beta * buf[1]; sum[2] = beta * buf[2]; }
            else { sum[0] += beta * buf[0]; sum[1] += beta * buf[1]; sum[2] += beta * buf[2]; }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int v = fabsf(sum[c]) < 2147483648.f ? (int)rintf(sum[c]) : (int)0x80000000;
            px[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
    }
    a.gray[(size_t)dy * a.gray_step + dx] = (uint8_t)((px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + (1 << 13)) >> 14);
}

// computeResizeAreaTab (resize.cpp), grouped by destination index
static void area_tab(int ssize, int dsize, double scale, std::vector<int>& start, std::vector<int>& si,
                     std::vector<float>& alpha) {
    start.assign(dsize + 1, 0); si.clear(); alpha.clear();
    for (int dx = 0; dx < dsize; dx++) {
        start[dx] = (int)si.size();
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) { si.push_back(sx1 - 1); alpha.push_back((float)((sx1 - fsx1) / cell)); }
        for (int sx = sx1; sx < sx2; sx++) { si.push_back(sx); alpha.push_back((float)(1.0 / cell)); }
        if (fsx2 - sx2 > 1e-3) { si.push_back(sx2); alpha.push_back((float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)); }
    }
    start[dsize] = (int)si.size();
}

extern "C" int rcflow_resize_area_bgr_to_gray_dev(rc_ctx* ctx, int stream, const uint8_t* d_bgr, size_t step, int sw,
                                                  int sh, uint8_t* d_gray, size_t gray_step, int dw, int dh) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!d_bgr || !d_gray || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || step < (size_t)sw * 3 || gray_step < (size_t)dw) {
        rc_set_error("bad frame arguments");
        return RC_EINVAL;
    }
    if (dw > sw || dh > sh) { rc_set_error("INTER_AREA is implemented for shrinking only (the reference's use)"); return RC_EINVAL; }
    RC_HIP(hipSetDevice(ctx->device));
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    const int isx = (int)nearbyint(scale_x), isy = (int)nearbyint(scale_y);
    RcAreaArgs a;
    memset(&a, 0, sizeof(a));
    a.bgr = d_bgr; a.step = step; a.sw = sw; a.sh = sh; a.dw = dw; a.dh = dh;
    a.gray = d_gray; a.gray_step = gray_step;
    a.fast = (fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON) ? 1 : 0;
    a.iscale_x = isx; a.iscale_y = isy;
    if (!a.fast) {
        std::vector<int> xs, xi, ys, yi;
        std::vector<float> xa, ya;
        area_tab(sw, dw, scale_x, xs, xi, xa);
        area_tab(sh, dh, scale_y, ys, yi, ya);
        const size_t nx = xi.size(), ny = yi.size();
        const size_t bytes = 4 * ((size_t)(dw + 1) + 2 * nx + (size_t)(dh + 1) + 2 * ny);
        int rc = rc_buf_ensure(s->area_tab, bytes);
        if (rc) return rc;
        int* base = (int*)s->area_tab.p;
        int* d_xs = base; int* d_xi = d_xs + (dw + 1); float* d_xa = (float*)(d_xi + nx);
        int* d_ys = (int*)(d_xa + nx); int* d_yi = d_ys + (dh + 1); float* d_ya = (float*)(d_yi + ny);
        // the tables are tiny; the blocking copies also order them before the launch on any stream
        RC_HIP(hipStreamSynchronize(s->cur));      // a previous launch may still read the old tables
        RC_HIP(hipMemcpy(d_xs, xs.data(), 4 * (dw + 1), hipMemcpyHostToDevice));
        RC_HIP(hipMemcpy(d_xi, xi.data(), 4 * nx, hipMemcpyHostToDevice));
        RC_HIP(hipMemcpy(d_xa, xa.data(), 4 * nx, hipMemcpyHostToDevice));
        RC_HIP(hipMemcpy(d_ys, ys.data(), 4 * (dh + 1), hipMemcpyHostToDevice));
        RC_HIP(hipMemcpy(d_yi, yi.data(), 4 * ny, hipMemcpyHostToDevice));
        RC_HIP(hipMemcpy(d_ya, ya.data(), 4 * ny, hipMemcpyHostToDevice));
        a.xstart = d_xs; a.xsi = d_xi; a.xalpha = d_xa; a.ystart = d_ys; a.ysi = d_yi; a.yalpha = d_ya;
    }
    {
        RcProfScope ps(ctx, s->cur, RC_K_PREPROC, 0, 3. * sw * sh + 1. * dw * dh);
        hipLaunchKernelGGL(k_resize_area_bgr_to_gray, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(RC_BLOCK), 0, s->cur, a);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}

// create_output(subframe, outmask) ripcurrents_module.cpp:225-244 (ripcurrents.cpp:487-505): the frame the
// reference writes to its "borders" video -- red channel forced to 255 wherever the edge mask is set.
__global__ __launch_bounds__(RC_BLOCK) void k_create_output(uint8_t* bgr, size_t step, const uint8_t* mask, size_t mask_step,
                                                            int w, int h) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    if (mask[(size_t)y * mask_step + x]) bgr[(size_t)y * step + 3 * x + 2] = 255;
}

extern "C" int rcflow_create_output_dev(rc_ctx* ctx, int stream, uint8_t* d_subframe_bgr, size_t step,
                                        const uint8_t* d_outmask, size_t mask_step, int w, int h) {
    RcSlot* s = rc_slot(ctx, stream);
    if (!s) return RC_EINVAL;
    if (!d_subframe_bgr || !d_outmask || w <= 0 || h <= 0 || step < (size_t)w * 3 || mask_step < (size_t)w) {
        rc_set_error("bad frame / mask arguments");
        return RC_EINVAL;
    }
    RC_HIP(hipSetDevice(ctx->device));
    {
        RcProfScope ps(ctx, s->cur, RC_K_OVERLAY, 0, 7. * w * h);
        hipLaunchKernelGGL(k_create_output, dim3((w + 63) / 64, (h + 3) / 4), dim3(RC_BLOCK), 0, s->cur, d_subframe_bgr, step,
                           d_outmask, mask_step, w, h);
    }
    RC_HIP(hipGetLastError());
    return RC_OK;
}
