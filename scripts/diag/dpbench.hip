// dpbench.hip -- issue rates of the fp64-side VALU instructions the exact expansion is made of (gfx950):
// cycles per wave64 instruction on one SIMD with W waves resident, for v_cvt_f64_f32, v_mul_f64, v_add_f64, v_fma_f64,
// v_cvt_f32_f64 and the fp32 v_fma_f32 / v_add_f32 for scale.
//   hipcc -O3 --offload-arch=gfx950 dpbench.hip -o dpbench && ./dpbench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096, CH = 8;
template <int OP>
__global__ void k(float* out, float seed) {
    float f[CH]; double d[CH];
    for (int i = 0; i < CH; i++) { f[i] = seed + i + threadIdx.x; d[i] = seed * 0.5 + i; }
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            if (OP == 1) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(d[(i + 1) % CH]));
            if (OP == 2) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(d[(i + 1) % CH]));
            if (OP == 3) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(d[(i + 1) % CH]), "v"(d[(i + 2) % CH]));
            if (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            if (OP == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(f[(i + 1) % CH]), "v"(f[(i + 2) % CH]));
            if (OP == 6) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[i]), "v"(f[(i + 1) % CH]));
            if (OP == 7) asm volatile("v_lshlrev_b32 %0, 3, %1\n\tv_add_u32 %0, %0, %1" : "=&v"(f[i]) : "v"(f[(i + 1) % CH]));
            if (OP == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(d[(i + 1) % CH]), "v"(d[(i + 2) % CH]));
            if (OP == 9) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(d[(i + 1) % CH]));
            if (OP == 10) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(d[(i + 1) % CH]));
            if (OP == 11) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[i]), "v"(f[(i + 1) % CH]));
        }
    }
    float acc = 0;
    for (int i = 0; i < CH; i++) acc += f[i] + (float)d[i];
    if (acc == 12345.f) out[0] = acc;
}
template <int OP>
int run(const char* name, int per_iter) {
    float* o; CHECK(hipMalloc(&o, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int dev; hipDeviceProp_t pr; CHECK(hipGetDevice(&dev)); CHECK(hipGetDeviceProperties(&pr, dev));
    const double ghz = pr.clockRate * 1e-6;
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd;           // one block per CU, `waves_per_simd` waves on each SIMD
        hipLaunchKernelGGL(k<OP>, dim3(pr.multiProcessorCount), dim3(threads), 0, 0, o, 1.f);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(pr.multiProcessorCount), dim3(threads), 0, 0, o, 1.f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc = ms * 1e-3 * ghz * 1e9;
        printf("%-16s %d waves/SIMD: %.2f cycles per wave-instruction per SIMD (nominal %.2f GHz)\n", name, waves_per_simd,
               cyc / ((double)ITER * CH * per_iter * waves_per_simd), ghz);
    }
    return 0;
}
int main() {
    run<0>("v_cvt_f64_f32", 1); run<1>("v_mul_f64", 1); run<2>("v_add_f64", 1); run<3>("v_fma_f64", 1);
    run<4>("v_cvt_f32_f64", 1); run<5>("v_fma_f32", 1); run<6>("v_add_f32", 1); run<7>("2 x int op", 2);
    run<8>("v_pk_fma_f32", 1); run<9>("v_pk_mul_f32", 1); run<10>("v_pk_add_f32", 1); run<11>("v_mul_f32", 1);
    return 0;
}
