// prints what __builtin_amdgcn_permlane16_swap / permlane32_swap return per lane (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    const unsigned lane = threadIdx.x;
    const auto s = __builtin_amdgcn_permlane16_swap(1000u + lane, 2000u + lane, false, false);
    o[lane] = s[0];
    o[64 + lane] = s[1];
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; ++r) printf("row %d lane %2d: ret0 = %u  ret1 = %u\n", r, 16 * r + 1, h[16 * r + 1], h[64 + 16 * r + 1]);
    return 0;
}
