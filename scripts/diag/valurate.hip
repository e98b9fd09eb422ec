// Vector-instruction issue rate on one SIMD of MI355X: cycles per wave64 instruction for a few instruction kinds, against
// the number of waves resident on the SIMD.  Each wave runs ITER loops of 32 independent instructions of the kind (eight
// register chains, so a wave alone is never waiting on its own previous result); s_memtime around the loop gives the wave's
// cycles, the launch's duration gives the SIMD's.  Prints cycles per instruction PER SIMD (= launch cycles / instructions a
// SIMD issued): the figure a kernel bound by vector issue has to be priced against.
//   hipcc -O3 --offload-arch=gfx950 valurate.hip -o valurate && ./valurate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { K_FMA, K_PKFMA, K_ADD, K_MULU24, K_MADU24, K_MULLO, K_ADDU, K_LSHLADD, K_CVT, K_DPP, K_RCP, K_FMA_DEP, K_MIX, K_FMA64, K_N };
static const char* kname[K_N] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_lo_u32", "v_add_u32",
                                 "v_lshl_add_u32", "v_cvt_f32_i32", "v_add_f32 dpp row_shr:1", "v_rcp_f32", "v_fma_f32 (one chain)",
                                 "fma+add_u32 alternating", "v_fma_f64"};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ __launch_bounds__(1024) void k_rate(int iters, float* out, long long* cyc) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f, b = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, p4 = {r1, r0}, p5 = {r3, r2}, p6 = {r5, r4}, p7 = {r7, r6}, pa = {a, a}, pb = {b, b};
    double d0 = r0, d1 = r1, d2 = r2, d3 = r3, d4 = r4, d5 = r5, d6 = r6, d7 = r7, da = 1.0001, db = 0.5;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7, ia = 3;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (KIND == K_FMA) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if (KIND == K_PKFMA) {
#define X(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##n) : "v"(pa), "v"(pb));
                REP8(X)
#undef X
            } else if (KIND == K_ADD) {
#define X(n) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r##n) : "v"(b));
                REP8(X)
#undef X
            } else if (KIND == K_MULU24) {
#define X(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(i##n) : "v"(ia));
                REP8(X)
#undef X
            } else if (KIND == K_MADU24) {
#define X(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i##n) : "v"(ia));
                REP8(X)
#undef X
            } else if (KIND == K_MULLO) {
#define X(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i##n) : "v"(ia));
                REP8(X)
#undef X
            } else if (KIND == K_ADDU) {
#define X(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i##n) : "v"(ia));
                REP8(X)
#undef X
            } else if (KIND == K_LSHLADD) {
#define X(n) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(i##n) : "v"(ia));
                REP8(X)
#undef X
            } else if (KIND == K_CVT) {
#define X(n) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(r##n) : "v"(i##n));
                REP8(X)
#undef X
            } else if (KIND == K_DPP) {
#define X(n) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r##n));
                REP8(X)
#undef X
            } else if (KIND == K_RCP) {
#define X(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(r##n));
                REP8(X)
#undef X
            } else if (KIND == K_FMA_DEP) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if (KIND == K_MIX) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_add_u32 %1, %1, %4" : "+v"(r##n), "+v"(i##n) : "v"(a), "v"(b), "v"(ia));
                X(0) X(1) X(2) X(3)
#undef X
            } else if (KIND == K_FMA64) {
#define X(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##n) : "v"(da), "v"(db));
                REP8(X)
#undef X
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y +
              (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7) + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
static void run(int wps, int iters, float* d_out, long long* d_cyc, double ghz, bool json) {
    // one block per CU with 4*wps waves: wps waves on each SIMD (eight: two blocks of 1024 threads per CU)
    const int threads = 64 * 4 * (wps > 4 ? 4 : wps), blocks = 256 * (wps > 4 ? wps / 4 : 1);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_rate<KIND><<<blocks, threads>>>(iters / 8, d_out, d_cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_rate<KIND><<<blocks, threads>>>(iters, d_out, d_cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long wc; CK(hipMemcpy(&wc, d_cyc, 8, hipMemcpyDeviceToHost));
    const double instr_per_simd = (double)iters * 32 * wps;      // the instrumented kind; K_MIX issues 32 of them too (16 + 16)
    const double launch_cycles = ms * 1e-3 * ghz * 1e9;
    printf(json ? "{\"kind\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_simd\": %.2f, \"wave_cycles_per_instr\": %.2f}\n"
                : "%-28s waves/SIMD %d : %5.2f cycles per instruction per SIMD (launch), %5.2f wave-clock ticks per instruction of one wave\n",
           kname[KIND], wps, launch_cycles / instr_per_simd, (double)wc / ((double)iters * 32));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    bool json = argc > 1 && !strcmp(argv[1], "--json");
    float* d_out; long long* d_cyc;
    CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_cyc, 64));
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double ghz = khz * 1e-6;
    if (!json) printf("device clock %.3f GHz (launch cycles are priced at this clock; the wave clock of s_memtime is 100 MHz-based on some parts)\n", ghz);
    const int iters = 20000;
#define RUNALL(K) for (int w : {1, 2, 4, 8}) run<K>(w, iters, d_out, d_cyc, ghz, json);
    RUNALL(K_FMA) RUNALL(K_PKFMA) RUNALL(K_ADD) RUNALL(K_MULU24) RUNALL(K_MADU24) RUNALL(K_MULLO) RUNALL(K_ADDU) RUNALL(K_LSHLADD)
    RUNALL(K_CVT) RUNALL(K_DPP) RUNALL(K_RCP) RUNALL(K_FMA_DEP) RUNALL(K_MIX) RUNALL(K_FMA64)
    return 0;
}
