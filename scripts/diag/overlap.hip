// Does a block's store drain overlap other blocks' compute?  Model of the expansion kernel: every 512-thread block
// computes (a dependent-free FMA loop sized like the kernel's VALU work, with LDS traffic and barriers), then writes a
// 64 x 32 tile of a float4 + float plane.  (a) one tile per block, as the kernel is launched now; (b) persistent blocks
// (3 per CU) looping over tiles, LDS-only barriers, so a wave goes on to the next tile while its stores drain.
//   hipcc -O3 --offload-arch=gfx950 -o overlap overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tile_work(f4* RA, float* RB, const unsigned char* src, int w, int h, int tile, int tiles_x, int tiles_y,
                                          int iters, float* lds, bool raw_barrier, int do_store) {
    const int z = tile / (tiles_x * tiles_y), r = tile - z * tiles_x * tiles_y;
    const int tx0 = (r % tiles_x) * 64, ty0 = (r / tiles_x) * 32;
    const int tid = threadIdx.x;
    // "staging": one byte per thread from the frame, through LDS
    const int sx = min(tx0 + (tid & 63), w - 1), sy = min(ty0 + (tid >> 6) * 4, h - 1);
    float v = (float)src[(size_t)z * w * h + (size_t)sy * w + sx];
    lds[tid] = v;
    if (raw_barrier) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
    float a0 = lds[(tid + 1) & 511], a1 = a0 * 0.5f, a2 = a0 + 1.f, a3 = a0 - 1.f;
    for (int i = 0; i < iters; i++) {      // four independent chains: issue-bound, like the passes
        a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 0.9999f, 0.25f);
        a2 = __builtin_fmaf(a2, 1.0002f, -0.5f); a3 = __builtin_fmaf(a3, 0.9998f, 0.125f);
    }
    if (raw_barrier) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
    if (!do_store) { if (a0 + a1 + a2 + a3 == 12345.678f) RB[0] = a0; return; }
    const int x = tx0 + (tid & 63), y0 = ty0 + (tid >> 6) * 4;
    if (x < w) {
        const size_t base = (size_t)z * w * h;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int y = y0 + rr;
            if (y < h) {
                const size_t p = base + (size_t)y * w + x;
                const f4 o = {a0, a1, a2, a3 + rr};
                __builtin_nontemporal_store(o, RA + p); __builtin_nontemporal_store(a0, RB + p);
            }
        }
    }
}

__global__ __launch_bounds__(512) void k_one(f4* RA, float* RB, const unsigned char* src, int w, int h, int tiles_x, int tiles_y, int iters, int do_store) {
    __shared__ float lds[12800];     // 50 KB: the expansion kernel's footprint (3 blocks per CU)
    tile_work(RA, RB, src, w, h, blockIdx.x, tiles_x, tiles_y, iters, lds, false, do_store);
}
__global__ __launch_bounds__(512) void k_persist(f4* RA, float* RB, const unsigned char* src, int w, int h, int tiles_x, int tiles_y, int ntiles, int iters, int do_store) {
    __shared__ float lds[12800];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) tile_work(RA, RB, src, w, h, t, tiles_x, tiles_y, iters, lds, true, do_store);
}

template <class F> static double timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e-3;
}

int main() {
    const int w = 1920, h = 1080, frames = 32, tx = 30, ty = 34, ntiles = tx * ty * frames;
    f4* RA; float* RB; unsigned char* src;
    CK(hipMalloc(&RA, (size_t)w * h * frames * 16)); CK(hipMalloc(&RB, (size_t)w * h * frames * 4)); CK(hipMalloc(&src, (size_t)w * h * frames));
    CK(hipMemset(src, 7, (size_t)w * h * frames));
    for (int iters : {0, 250, 500, 750, 1000}) {
        double c1 = timeit([&] { hipLaunchKernelGGL(k_one, dim3(ntiles), dim3(512), 0, 0, RA, RB, src, w, h, tx, ty, iters, 0); }, 5);
        double s1 = timeit([&] { hipLaunchKernelGGL(k_one, dim3(ntiles), dim3(512), 0, 0, RA, RB, src, w, h, tx, ty, iters, 1); }, 5);
        printf("iters %4d  one tile per block: compute only %.1f us, with stores %.1f us per frame", iters, c1 / frames * 1e6, s1 / frames * 1e6);
        for (int nb : {768, 1536}) {
            double sp = timeit([&] { hipLaunchKernelGGL(k_persist, dim3(nb), dim3(512), 0, 0, RA, RB, src, w, h, tx, ty, ntiles, iters, 1); }, 5);
            printf(" | persistent %d blocks: %.1f", nb, sp / frames * 1e6);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
