// rc_common.h -- shared declarations of the HIP implementation (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rcflow.h"

#define RC_MAX_LEVELS 12
#define RC_MAX_POLY_N 32      // taps -n..n kept on device
#define RC_MAX_WIN_M 32       // window radius winsize/2

// Built with -ffp-contract=off: a*b+c rounds twice unless written as fmaf().
#define RC_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

// `(int)float` as the reference's x86 build compiles it (cvttss2si): NaN and values outside int range give
// INT_MIN ("integer indefinite"), where v_cvt_i32_f32 gives 0 and saturates.  Bounds checks written
// for the x86 result (x < 1 || x + 2 > w) then reject such positions instead of letting them through.
__device__ __forceinline__ int rc_cvt_i32_x86(float v) {
    return (v >= -2147483648.f && v < 2147483648.f) ? (int)v : (int)0x80000000;
}

// Geometry and constants of one pyramid scale.
struct RcLevel {
    int w, h;            // level size (cvRound(W*scale))
    double scale_x;      // W0 / w   (resize.cpp scale_x)
    double scale_y;
    double sigma;        // pyramid blur sigma
    int ksize;           // pyramid blur taps
    int pyr_tw, pyr_th;  // pyr_level tile
    int pyr_reg_w, pyr_reg_h;  // LDS source region bounds (bytes / rows)
    size_t pyr_lds;
};

// Polynomial-expansion constants (FarnebackPrepareGaussian), taps 0..n.
struct RcPolyK {
    float g[RC_MAX_POLY_N + 1];
    float xg[RC_MAX_POLY_N + 1];
    float xxg[RC_MAX_POLY_N + 1];
    double ig11, ig03, ig33, ig55;
    double kdc;   // coefficient of the removed DC term in the yy/xx outputs
    int n;        // requested radius
    int n_eff;    // radius actually evaluated
};

// Window of FarnebackUpdateFlow_*: box (scale = 1/bs^2) or Gaussian (float taps).
struct RcWindow {
    float k[RC_MAX_WIN_M + 1];
    double box_scale;
    double box_eps;           // 1e-3 / box_scale^2: regulariser for unscaled window sums
    int m;
    int gaussian;
};

struct RcPyrArgs {
    const uint8_t* src;       // frame 0
    size_t src_step;          // bytes per row
    size_t src_frame_stride;  // bytes per frame
    int W0, H0;
    float* dst;               // I_k slot base
    size_t dst_slot_stride;   // floats per slot
    int dslot0, nslots, zstep; // destination slot of frame z is (dslot0 + z*zstep) % nslots
    int w, h;
    double scale_x, scale_y;
    int ksize;
    const float* kern;        // device, ksize floats
    int tw, th, reg_wp, reg_hmax;
    int direct;               // diagnostic (RC_ABL_PYR_STAGED): the earlier per-pixel / LDS-staged kernels
    int fixed3;               // scale 0: the fixed (1/4, 1/2, 1/4) taps of sigma <= 0 and an identity resize
};

// A pyramid scale written by the scale-0 expansion itself (k_polyexp<..., PYR = 1>)
struct RcPyrFused {
    float* dst;               // I_k slot base
    size_t dst_slot_stride;   // floats per slot
    int w, h;
    const float* kern;        // device: 3 taps (scale 1) / 9 taps (scale 2)
};

struct RcPolyArgs {
    const float* I;           // I_k slot base (scales >= 1, and the stage entry point)
    size_t I_slot_stride;
    const uint8_t* src8;      // scale 0: the 8-bit frames themselves (pyramid fused); else NULL
    size_t src8_step, src8_frame_stride;
    float4* RA;               // (y, x, yy, xx) coefficients
    float* RB;                // xy coefficient
    size_t R_slot_stride;     // elements per slot (pixels)
    int slot0, nslots, zstep;
    int w, h;
    int tile_h;               // rows per block: 32 or 48 (option "poly_tile_h")
    int no_fast_u8;           // diagnostic: byte-wise staging for every tile
    int valu_vertical;        // option "poly_mfma" = 0: vertical pass on the VALU instead of the matrix cores
    long long* stamps;        // diagnostic s_memtime stamps (RC_STAMPS builds; null in production)
    RcPolyK pk;
    int npyr;                 // scale 0 only: pyramid scales 1..npyr come out of this launch too
    RcPyrFused py[2];
};

// Option "ablate": earlier kernel forms and timing-only cuts kept for same-box A/B runs (0 in production).
// The first three change results (timing only); every other bit selects a bit-identical alternative.
enum RcAblate : int {
    RC_ABL_STAGE_A_ONLY = 1,         // winsize-3 kernels: stop after the first matrices stage
    RC_ABL_EXACT_PLAIN_SCANS = 2,    // option exact, box windows of winsize 3 / 5: the plain scans (V and G row-major in HBM, separate solve)
    RC_ABL_EXACT_FUSED_M = 16,       // option exact, box windows of winsize 3 / 5: FarnebackUpdateMatrices inside the column scan (measured slower)
    RC_ABL_NO_WINDOW = 4,            // skip window / solve / store
    RC_ABL_EMPTY_BLOCKS = 8,         // launch cost only
    RC_ABL_NO_FAST_U8 = 2048,        // expansion at scale 0: per-byte staging instead of dwords + v_perm
    RC_ABL_PYR_STAGED = 4096,        // pyramid: per-pixel / LDS-staged kernels
    RC_ABL_GENERIC_WINDOW = 8192,    // windows 5 / 10 / 20: runtime-sized generic kernel
    RC_ABL_BIG_32WIDE = 32768,       // winsize 20 tile kernel: 32-wide tiles
    RC_ABL_TILE_WINDOW = 65536,      // Gaussian winsize 10 / 20: tile kernel even for large launches
    RC_ABL_SWEEP_512T = 131072,      // strip-sweep kernel with 512 threads
    RC_ABL_FORCE_SWEEP = 8388608,    // strip-sweep kernel even for small launches
    RC_ABL_FORCE_CHAIN = 33554432,   // fused winsize-3 kernel: tile chains (option "chain") even for launches too small to want them
    RC_ABL_HIST_V1 = 16777216,       // histogram: the first form of the kernel (one scalar round per pixel, exact key everywhere)
};

struct RcIterArgs {
    const float4* RA;         // R slot base of this level
    const float* RB;
    size_t R_slot_stride;
    int slot0, slot1, nslots, zstep; // pair z uses slots (slot0 + z*zstep) % n and (slot1 + z*zstep) % n
    int w, h;
    // flow in: mode 0 zeros, 1 same-resolution, 2 coarser level (resize * mul)
    int in_mode;
    const float2* fin;
    size_t fin_pair_stride;   // elements
    int fin_w, fin_h;
    double up_scale_x, up_scale_y;
    float up_mul;
    int up_exact2;            // the coarse scale is exactly half-size (integer source coordinates, see k_flow_iter2_rr)
    // flow out
    char* fout;
    size_t fout_step;         // bytes per row
    size_t fout_pair_stride;  // bytes per pair
    int tw, th;               // tile (filled by the launcher)
    int tiles_x, tiles_y;
    int solve;                // 0: write flow_in (iterations == 0), 1: normal
    int xcd_remap;            // XCD-aware tile order (speed only)
    int ablate;               // timing-only ablation bits (0 in production)
    int addr32;               // every offset inside one frame's R planes and one pair's flow field fits 32 bits (set by the level driver)
    int chain_min_blocks;     // option "chain_min_blocks": a launch keeps at least this many blocks when its chains are shortened (0 = 4096)
    int chain;                // option "chain": consecutive pairs a block of the fused winsize-3 kernel walks on its tile (<= 1: none)
    RcWindow win;
};

// Pair groups of a chained launch (k_flow_iter2_rrc): block row g walks pairs start[g] .. start[g + 1] - 1.
#define RC_MAX_CHAIN_GROUPS 64
struct RcChainPlan {
    int ngroups;
    int start[RC_MAX_CHAIN_GROUPS + 1];
};

// Option "exact" (exact_kernels.hip): one scale of the upstream CPU operation order, staged through HBM.
struct RcExactArgs {
    const float4* RA;         // R slot base of this scale
    const float* RB;
    size_t n;                 // pixels of the scale (= R slot stride, plane stride of M / V)
    int slot0, slot1, nslots, zstep;
    int w, h;
    const float2* fin;        // flow of the coarser scale (null at the coarsest)
    size_t fin_pair_stride;
    int fin_w, fin_h;
    double up_scale_x, up_scale_y;
    float up_mul;
    float2* flow;             // [pairs][h][w]
    float* M;                 // [pairs][5][h][w]
    void* V;                  // [pairs][5][h][w]: float (Gaussian window) / double (box window)
    void* G;                  // box window: [pairs][5][h][w] double running row sums
    char* out;                // last iteration of scale 0: the caller's buffer (else null -> flow)
    size_t out_step, out_pair_stride;
    int plain_scans;          // RC_ABL_EXACT_PLAIN_SCANS
    int fused_matrices;       // RC_ABL_EXACT_FUSED_M
    RcWindow win;
};
void rc_launch_exact_polyexp(const RcPolyArgs& a, int frames, hipStream_t s);
void rc_launch_exact_flow_init(const RcExactArgs& a, int pairs, hipStream_t s);
void rc_launch_exact_matrices(const RcExactArgs& a, int pairs, hipStream_t s);
void rc_launch_exact_window_solve(const RcExactArgs& a, int pairs, hipStream_t s);
// box windows of winsize 3 / 5: matrices + window + solve of one iteration without M in HBM (reads a.flow, writes a.flow / a.out)
int rc_exact_iteration_fused_ok(const RcExactArgs& a);
void rc_launch_exact_iteration_fused(const RcExactArgs& a, int pairs, hipStream_t s);

// Raises a kernel's dynamic-LDS limit once per (call site, device): the attribute is per device, and a
// process may hold contexts on several devices.
#define RC_MAX_DEVICES 64
#define RC_ALLOW_LDS(fn, lds)                                                                             \
    do {                                                                                                  \
        static size_t rc_seen_[RC_MAX_DEVICES] = {0};                                                     \
        int rc_dev_ = 0;                                                                                  \
        (void)hipGetDevice(&rc_dev_);                                                                     \
        if (rc_dev_ >= 0 && rc_dev_ < RC_MAX_DEVICES && (size_t)(lds) > rc_seen_[rc_dev_]) {             \
            (void)hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds)); \
            rc_seen_[rc_dev_] = (size_t)(lds);                                                            \
        }                                                                                                 \
    } while (0)

void rc_launch_pyr(const RcPyrArgs& a, int frames, size_t lds, hipStream_t s);
void rc_launch_polyexp(const RcPolyArgs& a, int frames, hipStream_t s);
int rc_polyexp_pyr_ok(const RcPolyArgs& a);
// merged launches for a frame or two (see pyr_polyexp_kernels.hip)
int rc_pyr_pair_ok(const RcPyrArgs& a1, const RcPyrArgs& a2);
void rc_launch_pyr_pair(const RcPyrArgs& a1, const RcPyrArgs& a2, int frames, hipStream_t s);
int rc_polyexp_multi_ok(const RcPolyArgs* a, int nlev);
void rc_launch_polyexp_multi(const RcPolyArgs* a, int nlev, int frames, hipStream_t s);
// flow_iter_kernels.hip is built twice: fast arithmetic, and upstream's operation order (Gaussian windows)
#define RC_FLOW_DECLS                                                         \
    void rc_launch_flow_iter(const RcIterArgs& a, int pairs, hipStream_t s);  \
    /* two iterations in one launch (only where rc_flow_iter_can_fuse2 says so) */ \
    int rc_flow_iter_can_fuse2(const RcIterArgs& a);                          \
    int rc_flow_iter2_r0_reads(const RcIterArgs& a, int pairs);               \
    int rc_flow_chain_groups(const RcIterArgs& a, int pairs, int* starts, int cap); \
    void rc_launch_flow_iter2(const RcIterArgs& a, int pairs, hipStream_t s);
namespace rc_flow_fast { RC_FLOW_DECLS }
namespace rc_flow_exact { RC_FLOW_DECLS }

// interleave helpers for the stage-level test entry points
void rc_launch_pack_R5(const float* R5, float4* RA, float* RB, int n, hipStream_t s);
void rc_launch_unpack_R5(const float4* RA, const float* RB, float* R5, int n, hipStream_t s);
