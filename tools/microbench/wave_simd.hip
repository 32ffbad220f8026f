// Which SIMD each wave of a 512-thread workgroup lands on (HW_REG_HW_ID.SIMD_ID), one workgroup per CU.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/microbench/wave_simd.hip -o /tmp/wave_simd && /tmp/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

__global__ __launch_bounds__(512) void k(unsigned* out) {
    extern __shared__ unsigned char smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);   // HW_ID[15:0]
    const unsigned hw2 = __builtin_amdgcn_s_getreg((2 - 1) << 11 | 4 << 6 | 4);   // [5:4]
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw | hw2 << 16;
    if (threadIdx.x == 0) smem[0] = 1;
}

int main() {
    unsigned* d;
    const int blocks = 256;
    hipMalloc(&d, blocks * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    k<<<blocks, 512, 100 * 1024>>>(d);
    std::vector<unsigned> h(blocks * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<std::string, int> patterns;
    for (int b = 0; b < blocks; ++b) {
        std::string s;
        for (int w = 0; w < 8; ++w) s += std::to_string((h[b * 8 + w] >> 4) & 3) + (w == 7 ? "" : ",");
        patterns[s]++;
    }
    printf("SIMD_ID (HW_ID[5:4]) of waves 0..7 of a 512-thread workgroup, by frequency over %d workgroups:\n", blocks);
    for (auto& kv : patterns) printf("  %s  x %d\n", kv.first.c_str(), kv.second);
    printf("first workgroup raw HW_ID[15:0]:");
    for (int w = 0; w < 8; ++w) printf(" %04x(%u)", h[w] & 0xffff, h[w] >> 16);
    printf("\n");
    return 0;
}
