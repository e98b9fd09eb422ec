#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned d0 = 0x03020100u, d1 = 0x13121110u;
    o[0] = __builtin_amdgcn_perm(d0, d1, 0x0c010c07u);
    o[1] = __builtin_amdgcn_perm(d0, d1, 0x07060504u);
    o[2] = __builtin_amdgcn_perm(d0, d1, 0x03020100u);
    o[3] = __builtin_amdgcn_perm(d0, d1, 0x0c0c0c0cu);
}
int main() {
    unsigned* d; hipMalloc(&d, 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d);
    unsigned h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%08x %08x %08x %08x\n", h[0], h[1], h[2], h[3]);
    return 0;
}
