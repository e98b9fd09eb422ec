// LDS access-pattern microbenchmark (gfx950): bytes per clock per CU for the read/write patterns
// the polynomial-expansion and flow kernels use.  hipcc --offload-arch=gfx950 -O3 ldsbench.hip -o ldsbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NT = 512, ITERS = 2000, LDSF = 12 * 1024;   // 48 KB of floats

template <int MODE, int K = 0>
__global__ __launch_bounds__(NT) void k(float* out, int stride, int per_row, int lstride = 1) {
    __shared__ __align__(16) float s[LDSF];
    for (int i = threadIdx.x; i < LDSF; i += NT) s[i] = (float)i;
    __syncthreads();
    const int tid = threadIdx.x;
    const int row = tid / per_row, col = tid - row * per_row;
    float acc = 0.f;
    if (MODE == 0) {            // ds_read_b128: float4 index = row * stride/4 + col
        const float4* p = (const float4*)s + ((row * (stride / 4) + col * lstride) % (LDSF / 4));
        for (int it = 0; it < ITERS; it++) {
            float4 v = p[(it & 3) * 0];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            acc += v.x + v.w;
            asm volatile("" ::: "memory");
        }
    } else if (MODE == 1) {     // ds_read_b32
        const float* p = s + (row * stride + col * lstride) % LDSF;
        for (int it = 0; it < ITERS; it++) {
            float v = *p;
            asm volatile("" : "+v"(v));
            acc += v;
            asm volatile("" ::: "memory");
        }
    } else if (MODE == 2) {     // ds_read_b64
        const float2* p = (const float2*)s + (row * (stride / 2) + col * lstride) % (LDSF / 2);
        for (int it = 0; it < ITERS; it++) {
            float2 v = *p;
            asm volatile("" : "+v"(v.x), "+v"(v.y));
            acc += v.x + v.y;
            asm volatile("" ::: "memory");
        }
    } else if (MODE == 3) {     // ds_write_b128
        float4* p = (float4*)s + (row * (stride / 4) + col);
        float4 v = make_float4(tid, 1, 2, 3);
        for (int it = 0; it < ITERS; it++) {
            *p = v;
            asm volatile("" ::: "memory");
        }
        acc = s[tid];
    } else if (MODE == 4) {     // 4 x ds_read_b128 at +0,+1,+2,+3 float4 (the sliding window of the horizontal pass)
        const float4* p = (const float4*)s + (row * (stride / 4) + col);
        for (int it = 0; it < ITERS / 4; it++) {
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            asm volatile("" : "+v"(a.x), "+v"(b.y), "+v"(c.z), "+v"(d.w));
            acc += a.x + b.y + c.z + d.w;
            asm volatile("" ::: "memory");
        }
    }
    if (MODE == 5) {            // ds_write_b32
        float* p = s + (row * stride + col * lstride) % LDSF;
        for (int it = 0; it < ITERS; it++) { *p = acc; asm volatile("" ::: "memory"); }
        acc = s[tid];
    } else if (MODE == 6) {     // ds_write_b64
        float2* p = (float2*)s + (row * (stride / 2) + col * lstride) % (LDSF / 2);
        for (int it = 0; it < ITERS; it++) { *p = make_float2(acc, 1.f); asm volatile("" ::: "memory"); }
        acc = s[tid];
    } else if (MODE == 7) {     // one b128 read + K independent FMAs per iteration (do VALU and LDS overlap?)
        const float4* p = (const float4*)s + tid % (LDSF / 4);
        float r[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        for (int it = 0; it < ITERS; it++) {
            float4 v = *p;
#pragma unroll
            for (int q = 0; q < K; q++) r[q & 7] = __builtin_fmaf(r[q & 7], 1.0001f, 0.5f);
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            acc += v.x;
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int q = 0; q < 8; q++) acc += r[q];
    } else if (MODE == 8) {     // K FMAs only
        float r[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int q = 0; q < K; q++) r[q & 7] = __builtin_fmaf(r[q & 7], 1.0001f, 0.5f);
            asm volatile("" : "+v"(r[0]));
        }
#pragma unroll
        for (int q = 0; q < 8; q++) acc += r[q];
    } else if (MODE == 9) {     // one b32 read + K FMAs
        const float* p = s + tid;
        float r[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        for (int it = 0; it < ITERS; it++) {
            float v = *p;
#pragma unroll
            for (int q = 0; q < K; q++) r[q & 7] = __builtin_fmaf(r[q & 7], 1.0001f, 0.5f);
            asm volatile("" : "+v"(v));
            acc += v;
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int q = 0; q < 8; q++) acc += r[q];
    }
    if (acc == -1.2345f) out[tid] = acc;
}

template <int MODE, int K = 0>
static int run(const char* name, int bytes, int stride, int per_row, float* d, int lstride = 1) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 3 * 4;
    hipLaunchKernelGGL((k<MODE, K>), dim3(blocks), dim3(NT), 0, 0, d, stride, per_row, lstride);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, K>), dim3(blocks), dim3(NT), 0, 0, d, stride, per_row, lstride);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double total = (double)blocks * NT * ITERS * bytes;
    int clk = 0; CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    double bpc = total / (ms * 1e-3) / 256.0 / (clk * 1e3);
    printf("%-40s stride %3d per_row %3d lstride %2d: %7.3f ms  %6.1f B/clk/CU  %5.2f clk/wave-iter/CU\n", name, stride, per_row, lstride, ms, bpc, ms * 1e-3 * clk * 1e3 / ((double)blocks / 256 * (NT / 64) * ITERS));
    return 0;
}

int main() {
    float* d; CK(hipMalloc(&d, 4096));
    run<1>("b32 contiguous", 4, 64, 64, d);
    run<2>("b64 contiguous", 8, 128, 64, d);
    run<0>("b128 contiguous", 16, 256, 64, d);
    for (int pitch : {64, 72, 80, 84, 88, 96, 112, 128}) run<0>("b128 16 lanes per row", 16, pitch, 16, d);
    for (int ls : {1, 2, 3, 5, 13, 17, 21}) run<0>("b128 lane stride (float4 units)", 16, 0, 64, d, ls);
    for (int ls : {1, 2, 3, 5, 13, 26, 33}) run<2>("b64 lane stride (float2 units)", 8, 0, 64, d, ls);
    for (int ls : {1, 2, 3, 33, 52, 65}) run<1>("b32 lane stride (floats)", 4, 0, 64, d, ls);
    run<3>("write b128 contiguous", 16, 256, 64, d);
    run<6>("write b64 contiguous", 8, 128, 64, d);
    run<5>("write b32 contiguous", 4, 64, 64, d);
    for (int ls : {2, 33, 52, 65}) run<5>("write b32 lane stride", 4, 0, 64, d, ls);
    run<8, 8>("8 FMA only", 0, 0, 64, d);
    run<8, 32>("32 FMA only", 0, 0, 64, d);
    run<7, 0>("b128 read + 0 FMA", 16, 0, 64, d);
    run<7, 4>("b128 read + 4 FMA", 16, 0, 64, d);
    run<7, 8>("b128 read + 8 FMA", 16, 0, 64, d);
    run<7, 16>("b128 read + 16 FMA", 16, 0, 64, d);
    run<7, 32>("b128 read + 32 FMA", 16, 0, 64, d);
    run<9, 0>("b32 read + 0 FMA", 4, 0, 64, d);
    run<9, 4>("b32 read + 4 FMA", 4, 0, 64, d);
    run<9, 8>("b32 read + 8 FMA", 4, 0, 64, d);
    run<9, 16>("b32 read + 16 FMA", 4, 0, 64, d);
    return 0;
}
