// Store-pattern microbenchmark for the expansion kernel's outputs: every block writes a TW x TH tile of a
// [frames][h][w] float4 plane (RA) and float plane (RB), nontemporal, no computation.  Which tile shape does the
// memory system take fastest?   hipcc -O3 --offload-arch=gfx950 -o tilestore tilestore.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int TW, int TH, int NT, int NTEMP>
__global__ __launch_bounds__(NT) void k_store(f4* RA, float* RB, int w, int h) {
    constexpr int COLS = TW, ROWS_PER_PASS = NT / COLS, NR = TH / ROWS_PER_PASS;
    const int x = blockIdx.x * TW + (threadIdx.x % COLS), y0 = blockIdx.y * TH + (threadIdx.x / COLS) * NR;
    if (x >= w) return;
    const size_t base = (size_t)blockIdx.z * w * h;
    const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int y = y0 + r;
        if (y < h) {
            const size_t p = base + (size_t)y * w + x;
            if (NTEMP) { __builtin_nontemporal_store(v, RA + p); __builtin_nontemporal_store(v.w, RB + p); }
            else { RA[p] = v; RB[p] = v.w; }
        }
    }
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

template <int TW, int TH, int NT>
static void run(f4* RA, float* RB, int w, int h, int frames, const char* name) {
    dim3 grid((w + TW - 1) / TW, (h + TH - 1) / TH, frames);
    const double bytes = 20.0 * w * h * frames;
    double t1 = timeit([&] { hipLaunchKernelGGL((k_store<TW, TH, NT, 1>), grid, dim3(NT), 0, 0, RA, RB, w, h); }, 10);
    double t0 = timeit([&] { hipLaunchKernelGGL((k_store<TW, TH, NT, 0>), grid, dim3(NT), 0, 0, RA, RB, w, h); }, 10);
    printf("%-28s nontemporal %.2f TB/s (%.2f us per frame)   plain %.2f TB/s\n", name, bytes / t1 * 1e-12, t1 / frames * 1e6, bytes / t0 * 1e-12);
    fflush(stdout);
}

int main() {
    const int w = 1920, h = 1080, frames = 32;
    f4* RA; float* RB;
    CK(hipMalloc(&RA, (size_t)w * h * frames * 16)); CK(hipMalloc(&RB, (size_t)w * h * frames * 4));
    for (int rep = 0; rep < 2; rep++) {
        run<64, 32, 512>(RA, RB, w, h, frames, "64x32 tile, 512 thr (now)");
        run<64, 32, 256>(RA, RB, w, h, frames, "64x32 tile, 256 thr");
        run<128, 16, 512>(RA, RB, w, h, frames, "128x16 tile, 512 thr");
        run<256, 8, 512>(RA, RB, w, h, frames, "256x8 tile, 512 thr");
        run<64, 64, 512>(RA, RB, w, h, frames, "64x64 tile, 512 thr");
        run<128, 32, 512>(RA, RB, w, h, frames, "128x32 tile, 512 thr");
        run<64, 16, 256>(RA, RB, w, h, frames, "64x16 tile, 256 thr");
        run<32, 32, 256>(RA, RB, w, h, frames, "32x32 tile, 256 thr");
        run<64, 8, 512>(RA, RB, w, h, frames, "64x8 tile, 512 thr (1 row/thr)");
        run<512, 4, 512>(RA, RB, w, h, frames, "512x4 tile, 512 thr");
    }
    return 0;
}
