// Which physical CUs does bit i of a HIP stream's CU mask enable on MI355X?   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
// For a few bit patterns: a stream created with hipExtStreamCreateWithCUMask runs a grid of whole-CU workgroups (1024 threads, 64 KB LDS) that each note
// HW_REG_XCC_ID and the CU / SH / SE fields of HW_REG_HW_ID; the host prints how many distinct CUs each XCD contributed and how long the grid took
// against the same grid on an unmasked stream.  (Stand-alone hardware probe: no library code involved.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>
#include <string>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_where(unsigned* out, int spin) {
    __shared__ unsigned pad[16384];                       // 64 KB: together with 1024 threads at most two workgroups per CU
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned acc = pad[(threadIdx.x * 7) & 16383];
    const long long t0 = clock64();
    while (clock64() - t0 < spin) acc += pad[(acc + threadIdx.x) & 16383];        // keep the CU busy for a while so that the grid spreads
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = (xcc & 0xf) | (acc & 0x80000000u & 0); out[2 * blockIdx.x + 1] = hw; }
}

// a "persistent" grid: one whole-CU workgroup (1024 threads, 128 KB LDS) per CU of the mask, each busy for a fixed time: one round if every workgroup finds a CU of its own
__global__ __launch_bounds__(1024) void k_fixed(unsigned* out, int spin) {
    extern __shared__ unsigned big[];
    big[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned acc = big[(threadIdx.x * 7) & 1023];
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) acc += big[(acc + threadIdx.x) & 1023];
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
static void run_fixed(const char* name, const std::vector<uint32_t>& mask, bool masked, unsigned* d_out, int blocks) {
    hipStream_t s;
    if (masked) { if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed\n", name); return; } }
    else (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fixed), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_fixed, dim3(blocks), dim3(1024), 128 * 1024, s, d_out, 1000);
    (void)hipStreamSynchronize(s);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_fixed, dim3(blocks), dim3(1024), 128 * 1024, s, d_out, 10000);      // 10,000 ticks of the 100 MHz counter = 100 us per workgroup
        (void)hipEventRecord(e1, s);
        (void)hipStreamSynchronize(s);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("fixed-work grid  %-34s %3d whole-CU workgroups: %.3f ms\n", name, blocks, best);
    (void)hipStreamDestroy(s);
}

static void run(const char* name, const std::vector<uint32_t>& mask, bool masked, unsigned* d_out, int blocks) {
    hipStream_t s;
    if (masked) { if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed\n", name); return; } }
    else (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(1024), 0, s, d_out, 1000);
    (void)hipStreamSynchronize(s);
    (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(1024), 0, s, d_out, 200000);
    (void)hipEventRecord(e1, s);
    (void)hipStreamSynchronize(s);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(2 * blocks);
    (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::set<unsigned> per_xcc[16];
    for (int b = 0; b < blocks; ++b) per_xcc[h[2 * b] & 0xf].insert((h[2 * b + 1] >> 8) & 0xff);      // CU_ID[11:8] | SH_ID[12] | SE_ID[15:13]
    int total = 0;
    std::string line;
    for (int x = 0; x < 8; ++x) { total += (int)per_xcc[x].size(); line += " " + std::to_string(per_xcc[x].size()); }
    printf("%-44s distinct CUs %3d  per XCD:%s   %d blocks in %.3f ms\n", name, total, line.c_str(), blocks, ms);
    (void)hipStreamDestroy(s);
}

int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    const int blocks = 2048;
    unsigned* d_out;
    CHK(hipMalloc((void**)&d_out, 2 * blocks * sizeof(unsigned)));
    auto bits = [](int lo, int hi, int stride = 1, int phase = 0) { std::vector<uint32_t> m(8, 0); for (int i = lo; i < hi; ++i) if (i % stride == phase || stride == 1) m[i >> 5] |= 1u << (i & 31); return m; };
    run("no mask", bits(0, 256), false, d_out, blocks);
    run("bits 0..255", bits(0, 256), true, d_out, blocks);
    run("bits 0..127", bits(0, 128), true, d_out, blocks);
    run("bits 128..255", bits(128, 256), true, d_out, blocks);
    run("bits 0..87", bits(0, 88), true, d_out, blocks);
    run("bits 88..167", bits(88, 168), true, d_out, blocks);
    run("bits 168..255", bits(168, 256), true, d_out, blocks);
    run("bits 0..31", bits(0, 32), true, d_out, blocks);
    run("bits 0..7", bits(0, 8), true, d_out, blocks);
    run("bits i % 8 == 0", bits(0, 256, 8, 0), true, d_out, blocks);
    run("bits i % 8 == 3", bits(0, 256, 8, 3), true, d_out, blocks);
    run("bits i % 32 == 0", bits(0, 256, 32, 0), true, d_out, blocks);
    {   // how long is one workgroup alone?  (clock64 may count the shader clock or the 100 MHz reference: scale everything by this)
        run_fixed("no mask, 1 workgroup", bits(0, 256), false, d_out, 1);
        run_fixed("no mask", bits(0, 256), false, d_out, 256);
        run_fixed("no mask", bits(0, 256), false, d_out, 128);
        run_fixed("bits 0..127", bits(0, 128), true, d_out, 128);
        run_fixed("bits 0..95", bits(0, 96), true, d_out, 96);
        run_fixed("bits 96..159", bits(96, 160), true, d_out, 64);
        run_fixed("bits 160..255", bits(160, 256), true, d_out, 96);
        run_fixed("bits 0..87", bits(0, 88), true, d_out, 88);
        run_fixed("bits 0..87", bits(0, 88), true, d_out, 64);
        run_fixed("bits 88..167", bits(88, 168), true, d_out, 80);
        run_fixed("bits 0..63", bits(0, 64), true, d_out, 64);
        run_fixed("bits 0..31", bits(0, 32), true, d_out, 32);
    }
    // two masked streams at once: do disjoint masks run side by side?
    {
        hipStream_t a, b;
        auto ma = bits(0, 128), mb = bits(128, 256);
        if (hipExtStreamCreateWithCUMask(&a, 8, ma.data()) == hipSuccess && hipExtStreamCreateWithCUMask(&b, 8, mb.data()) == hipSuccess) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            unsigned* d2; (void)hipMalloc((void**)&d2, 2 * blocks * sizeof(unsigned));
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, a);
            hipLaunchKernelGGL(k_where, dim3(blocks), dim3(1024), 0, a, d_out, 200000);
            hipLaunchKernelGGL(k_where, dim3(blocks), dim3(1024), 0, b, d2, 200000);
            (void)hipStreamSynchronize(b);
            (void)hipEventRecord(e1, a);
            (void)hipStreamSynchronize(a);
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("two disjoint halves side by side: 2 x %d blocks in %.3f ms (one half alone: see 'bits 0..127')\n", blocks, ms);
        }
    }
    return 0;
}
