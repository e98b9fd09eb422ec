// Does a run of one-pass vector instructions keep its two-cycle rate inside a kernel shaped like the expansion's horizontal
// pass?  Block = 512 threads, 48 KB of LDS (three blocks per CU); per trip every thread reads 20 floats from LDS (five
// ds_read_b128), runs 4 pixels x 7 taps x (add, sub, 3 fused multiply-adds) on them and writes 12 floats back, then the block
// meets at a barrier.  MODE 0: the 22 filter coefficients in vector registers (every arithmetic instruction one-pass);
// MODE 1: in scalar registers (the products two-pass), as the compiler leaves kernel arguments.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off frun.hip -o frun && ./frun
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
struct Coef { float g[8], xg[8], xxg[8]; };
__device__ __forceinline__ float vreg(float x) { asm("" : "+v"(x)); return x; }
template <int MODE, int SYNC, int LDS_>
__global__ __launch_bounds__(512) void k(Coef c, int trips, float* out) {
    extern __shared__ __align__(16) float sm[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 12288; i += 512) sm[i] = (float)(i & 255) * 0.01f;
    __syncthreads();
    float gv[8], xgv[8], xxgv[8];
#pragma unroll
    for (int k_ = 0; k_ < 8; k_++) {
        gv[k_] = MODE == 0 ? vreg(c.g[k_]) : c.g[k_];
        xgv[k_] = MODE == 0 ? vreg(c.xg[k_]) : c.xg[k_];
        xxgv[k_] = MODE == 0 ? vreg(c.xxg[k_]) : c.xxg[k_];
    }
    float acc = 0.f;
    // (like the expansion's horizontal pass: a lane's window starts 16 bytes after its neighbour's; outputs 16 bytes apart, three planes)
    const float4* in = (const float4*)(sm + tid * 4);
    float4* o4 = (float4*)(sm + 4096 + tid * 4);
    for (int t = 0; t < trips; t++) {
        float v[20];
        if (LDS_ == 1 || LDS_ == 2) {
            asm volatile("" ::: "memory");         // (the reads are not loop invariants)
#pragma unroll
            for (int q = 0; q < 5; q++) { float4 x = in[q + 8 * (t & 1)]; v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; }
        } else {
#pragma unroll
            for (int q = 0; q < 20; q++) { v[q] = acc + (float)q; asm volatile("" : "+v"(v[q])); }
        }
        float h0[4], h1[4], h2[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int cc = 8 + p;
            float s0 = v[cc] * gv[0], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k_ = 1; k_ <= 7; k_++) {
                float su = v[cc + k_] + v[cc - k_], df = v[cc + k_] - v[cc - k_];
                s0 = __builtin_fmaf(su, gv[k_], s0);
                s1 = __builtin_fmaf(df, xgv[k_], s1);
                s2 = __builtin_fmaf(su, xxgv[k_], s2);
            }
            h0[p] = s0; h1[p] = s1; h2[p] = s2;
        }
        if (LDS_ == 1 || LDS_ == 3) {
            o4[0] = make_float4(h0[0], h0[1], h0[2], h0[3]);
            o4[512] = make_float4(h1[0], h1[1], h1[2], h1[3]);
            o4[1024] = make_float4(h2[0], h2[1], h2[2], h2[3]);
            acc += h0[0];
            if (LDS_ == 3) acc += h1[1] * 1e-9f;
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) acc += h0[p] + h1[p] + h2[p];
        }
        if (SYNC) __syncthreads();
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int MODE, int SYNC, int LDS_>
static void run(const char* name, int blocks_per_cu) {
    Coef c;
    for (int i = 0; i < 8; i++) { c.g[i] = 0.1f + 0.01f * i; c.xg[i] = 0.02f * i; c.xxg[i] = 0.003f * i * i; }
    float* out; CK(hipMalloc(&out, 64));
    const int trips = 2000;
    const size_t lds = 49152;
    CK(hipFuncSetAttribute((const void*)k<MODE, SYNC, LDS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE, SYNC, LDS_><<<256 * blocks_per_cu, 512, lds>>>(c, trips / 10, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<MODE, SYNC, LDS_><<<256 * blocks_per_cu, 512, lds>>>(c, trips, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // vector instructions of the arithmetic per trip and wave: 4 px x (1 + 7 x 5) = 144
    const double inst = 144.0, waves_per_simd = 2.0 * blocks_per_cu;
    printf("%-46s blocks/CU %d: %7.3f ms, %5.2f cycles per arithmetic instruction per SIMD (2.4 GHz)\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / (trips * inst * waves_per_simd));
}
int main() {
    for (int b : {1, 2, 3}) {
        run<0, 1, 1>("vector-register coefficients, LDS, barrier", b);
        run<1, 1, 1>("scalar-register coefficients, LDS, barrier", b);
        run<0, 0, 1>("vector-register coefficients, LDS, no barrier", b);
        run<1, 0, 1>("scalar-register coefficients, LDS, no barrier", b);
        run<0, 0, 2>("vector-register coefficients, LDS reads only, no barrier", b);
        run<0, 0, 3>("vector-register coefficients, LDS writes only, no barrier", b);
        run<0, 0, 0>("vector-register coefficients, registers only", b);
        run<1, 0, 0>("scalar-register coefficients, registers only", b);
    }
    return 0;
}
