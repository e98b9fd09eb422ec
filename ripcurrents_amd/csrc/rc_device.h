// rc_device.h -- device-side helpers shared by the gfx950 kernels.
#pragma once

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

