// clkmon.hip -- the shader clock the GPU actually runs at while a workload executes (diagnostic, not part of the library).
// One wave on its own stream, at wave priority 3, times a chain of 4096 dependent scalar additions against s_memrealtime
// (constant 100 MHz counter) every `period_us`: a lone wave issues one of them per fixed number of shader cycles, so the
// chain's duration is inversely proportional to the clock its CU runs at.  (s_memtime is no use for this: it advances at
// exactly 24 ticks per 100 MHz tick on this part, idle or loaded.)
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC clkmon.hip -o libclkmon.so     (scripts/r2/clock_under_load.py loads it)
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k_clkmon(long long* out, int n, long long period_ticks) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; i++) {                       // bounded: n samples of a wall-clock period each
        __builtin_amdgcn_s_setprio(3);
        const long long w0 = __builtin_amdgcn_s_memrealtime();
        int v = i;
        // 128 rounds of 32 dependent additions: 160 bytes of code (a 16 KB straight-line chain would measure the
        // workload's instruction-cache pressure as well)
        int cnt = 128;
        asm volatile("s_waitcnt lgkmcnt(0)\n"
                     "1:\n\t.rept 32\n\ts_add_u32 %0, %0, 1\n\t.endr\n\t"
                     "s_sub_u32 %1, %1, 1\n\ts_cmp_lg_u32 %1, 0\n\ts_cbranch_scc1 1b" : "+s"(v), "+s"(cnt) : : "scc");
        const long long w1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_setprio(0);
        out[2 * i] = w1 - w0;                           // 100 MHz ticks for 4096 dependent scalar instructions
        out[2 * i + 1] = v;
        while ((long long)__builtin_amdgcn_s_memrealtime() - w0 < period_ticks) __builtin_amdgcn_s_sleep(64);
    }
}
static hipStream_t g_s = nullptr;
static long long* g_d = nullptr;
static int g_n = 0;
extern "C" int clkmon_start(int samples, int period_us) {
    if (g_d) return -1;
    if (hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking) != hipSuccess) return -2;
    if (hipMalloc(&g_d, sizeof(long long) * 2 * samples) != hipSuccess) return -3;
    g_n = samples;
    hipLaunchKernelGGL(k_clkmon, dim3(1), dim3(64), 0, g_s, g_d, samples, (long long)period_us * 100);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
extern "C" int clkmon_read(long long* host_out) {      // blocks until the monitor has taken all its samples
    if (!g_d) return -1;
    if (hipStreamSynchronize(g_s) != hipSuccess) return -2;
    if (hipMemcpy(host_out, g_d, sizeof(long long) * 2 * g_n, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    (void)hipFree(g_d); (void)hipStreamDestroy(g_s);
    g_d = nullptr; g_s = nullptr;
    return g_n;
}

// A synthetic load for telling clock from issue contention: every SIMD of every CU holds 8 waves of independent fp32
// FMA chains with scalar work in between (no memory traffic).  If the monitor's chain keeps its duration beside this,
// what lengthens it beside a real workload is the clock, not other waves competing for its SIMD's issue slots.
__global__ __launch_bounds__(512) void k_burn(float* out, int iters, int salu) {
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = threadIdx.x * 0.001f + i;
    int sacc = blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
        if (salu) asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 3\n\ts_lshl_b32 %0, %0, 1\n\ts_add_u32 %0, %0, 7" : "+s"(sacc));
    }
    float acc = (float)sacc;
    for (int i = 0; i < 8; i++) acc += f[i];
    if (acc == 12345.f) out[0] = acc;
}
static float* g_burn = nullptr;
extern "C" int clkmon_burn(int iters, int salu) {       // asynchronous, on the null stream; 2 blocks of 512 per CU... x 4 rounds
    if (!g_burn && hipMalloc(&g_burn, 64) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_burn, dim3(256 * 4 * 4), dim3(512), 0, 0, g_burn, iters, salu);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
