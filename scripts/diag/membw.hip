// Memory-system microbenchmarks for MI355X: streaming read (float4 loads, LDS-DMA), write, copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, size_t n4, float* out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// contiguous chunk per block (tile-like access)
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read_chunk(const float4* __restrict__ p, size_t n4, float* out) {
    size_t per = n4 / gridDim.x;
    const float4* q = p + per * blockIdx.x;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < per; i += UNROLL * 256) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = q[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read_dma(const float4* __restrict__ p, size_t n4, float* out) {
    __shared__ float4 buf[8 * 256];
    size_t per = n4 / gridDim.x;
    const float4* q = p + per * blockIdx.x;
    const int wave_base = threadIdx.x & ~63;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i + 7 * 256 < per; i += 8 * 256) {
#pragma unroll
        for (int u = 0; u < 8; u++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(q + i + u * 256),
                                             (__attribute__((address_space(3))) void*)(buf + u * 256 + wave_base), 16, 0, 0);
        __syncthreads();
        acc += buf[threadIdx.x].x;
        __syncthreads();
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ p, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ s, float4* __restrict__ d, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) d[i] = s[i];
}

typedef float rc_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_write_nt(rc_f4* __restrict__ p, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const rc_f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(v, p + i);
}
__global__ __launch_bounds__(256) void k_copy_nt(const rc_f4* __restrict__ s, rc_f4* __restrict__ d, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}

// more bytes in flight per thread: U independent 16-byte loads before the stores (grid-stride)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_u(const rc_f4* __restrict__ s, rc_f4* __restrict__ d, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        rc_f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) d[i] = s[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_write_u(rc_f4* __restrict__ p, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const rc_f4 v = {1.f, 2.f, 3.f, 4.f};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride)
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v, p + i + u * stride); else p[i + u * stride] = v; }
    for (; i < n4; i += stride) p[i] = v;
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

#include <cstring>
#include <algorithm>
int main(int argc, char** argv) {
    const size_t bytes = (size_t)2 << 30;     // 2 GiB, well beyond the 256 MB Infinity Cache
    const size_t n4 = bytes / 16;
    float4 *src, *dst; float* out;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
    if (argc > 1 && !strcmp(argv[1], "--json")) {
        // one line for bench.py: the best streaming read / write / copy rate over launch shapes and variants, GB/s
        // (copy counts read + written bytes, like the guide's 6.29 TB/s float4 copy)
        double rd = 0, wr = 0, cp = 0;
        for (int rep = 0; rep < 2; rep++)
            for (int blocks : {2048, 4096, 8192, 16384}) {
                rd = std::max(rd, bytes / timeit([&] { hipLaunchKernelGGL(k_read<8>, dim3(blocks), dim3(256), 0, 0, src, n4, out); }, 5));
                rd = std::max(rd, bytes / timeit([&] { hipLaunchKernelGGL(k_read<4>, dim3(blocks), dim3(256), 0, 0, src, n4, out); }, 5));
                wr = std::max(wr, bytes / timeit([&] { hipLaunchKernelGGL((k_write_u<4, false>), dim3(blocks), dim3(256), 0, 0, (rc_f4*)dst, n4); }, 5));
                wr = std::max(wr, bytes / timeit([&] { hipLaunchKernelGGL((k_write_u<4, true>), dim3(blocks), dim3(256), 0, 0, (rc_f4*)dst, n4); }, 5));
                wr = std::max(wr, bytes / timeit([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, dst, n4); }, 5));
                cp = std::max(cp, 2.0 * bytes / timeit([&] { hipLaunchKernelGGL((k_copy_u<4, false>), dim3(blocks), dim3(256), 0, 0, (const rc_f4*)src, (rc_f4*)dst, n4); }, 5));
                cp = std::max(cp, 2.0 * bytes / timeit([&] { hipLaunchKernelGGL((k_copy_u<4, true>), dim3(blocks), dim3(256), 0, 0, (const rc_f4*)src, (rc_f4*)dst, n4); }, 5));
                cp = std::max(cp, 2.0 * bytes / timeit([&] { hipLaunchKernelGGL((k_copy_u<8, true>), dim3(blocks), dim3(256), 0, 0, (const rc_f4*)src, (rc_f4*)dst, n4); }, 5));
                cp = std::max(cp, 2.0 * bytes / timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, src, dst, n4); }, 5));
            }
        printf("{\"read\": %.1f, \"write\": %.1f, \"copy\": %.1f, \"unit\": \"GB/s\", \"tool\": \"scripts/diag/membw.hip: best of float4 grid-stride "
               "kernels (1-8 loads in flight per thread, plain and nontemporal, 2048-16384 blocks), 2 GiB buffers; copy = read + written bytes\"}\n",
               rd * 1e-9, wr * 1e-9, cp * 1e-9);
        return 0;
    }
    for (int rep = 0; rep < 2; rep++) {      // second pass = warm clocks
        for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
            double t1 = timeit([&] { hipLaunchKernelGGL(k_read<1>, dim3(blocks), dim3(256), 0, 0, src, n4, out); }, 10);
            double t4 = timeit([&] { hipLaunchKernelGGL(k_read<4>, dim3(blocks), dim3(256), 0, 0, src, n4, out); }, 10);
            double t8 = timeit([&] { hipLaunchKernelGGL(k_read<8>, dim3(blocks), dim3(256), 0, 0, src, n4, out); }, 10);
            double tc = timeit([&] { hipLaunchKernelGGL(k_read_chunk<8>, dim3(blocks), dim3(256), 0, 0, src, n4, out); }, 10);
            double td = timeit([&] { hipLaunchKernelGGL(k_read_dma, dim3(blocks), dim3(256), 0, 0, src, n4, out); }, 10);
            double tw = timeit([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, dst, n4); }, 10);
            double tcp = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, src, dst, n4); }, 10);
            double twn = timeit([&] { hipLaunchKernelGGL(k_write_nt, dim3(blocks), dim3(256), 0, 0, (rc_f4*)dst, n4); }, 10);
            double tcn = timeit([&] { hipLaunchKernelGGL(k_copy_nt, dim3(blocks), dim3(256), 0, 0, (const rc_f4*)src, (rc_f4*)dst, n4); }, 10);
            printf("blocks %6d  nontemporal: write %.2f  copy(r+w) %.2f TB/s\n", blocks, bytes / twn * 1e-12, 2.0 * bytes / tcn * 1e-12);
            printf("blocks %6d  read u1 %.2f  u4 %.2f  u8 %.2f  chunk u8 %.2f  lds-dma %.2f | write %.2f | copy(r+w) %.2f  TB/s\n", blocks,
                   bytes / t1 * 1e-12, bytes / t4 * 1e-12, bytes / t8 * 1e-12, bytes / tc * 1e-12, bytes / td * 1e-12,
                   bytes / tw * 1e-12, 2.0 * bytes / tcp * 1e-12);
            fflush(stdout);
        }
    }
    return 0;
}
