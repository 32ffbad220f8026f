// Where do the workgroups of a persistent launch land?  512 workgroups of 256 threads with 80 KiB of LDS each
// (conv_t32's two-per-CU tiles) record HW_ID, XCC_ID and their start time, then spin so that all stay resident.
// Prints, per CU, the block ids and TG_IDs of its workgroups: which pairs share a CU, and whether TG_ID parity
// tells them apart.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/microbench/wg_placement.hip -o /tmp/wgp && /tmp/wgp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);    // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20);  // HW_REG_XCC_ID
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)t;
        out[blockIdx.x * 4 + 3] = (unsigned)(t >> 32);
        smem[0] = 1;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}

int main() {
    const int G = 512;
    unsigned* d;
    hipMalloc(&d, G * 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        probe<<<G, 256, 81280>>>(d, 200);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(G * 4);
    hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cus;
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < G; ++b) t0 = std::min(t0, ((unsigned long long)h[b * 4 + 3] << 32) | h[b * 4 + 2]);
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus[xcc << 12 | se << 8 | sh << 4 | cu].push_back(b);
    }
    int hist[8] = {}, parity_ok = 0, pairs = 0;
    for (auto& kv : cus) {
        hist[std::min<size_t>(kv.second.size(), 7)]++;
        if (kv.second.size() == 2) {
            ++pairs;
            const unsigned a = (h[kv.second[0] * 4] >> 16) & 0xf, b = (h[kv.second[1] * 4] >> 16) & 0xf;
            parity_ok += (a & 1) != (b & 1);
        }
    }
    printf("CUs seen: %zu; workgroups per CU histogram:", cus.size());
    for (int i = 0; i < 8; ++i) printf(" %d:%d", i, hist[i]);
    printf("\npairs whose TG_IDs differ in parity: %d of %d\n", parity_ok, pairs);
    int shown = 0;
    for (auto& kv : cus) {
        if (shown++ >= 24) break;
        printf("xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 0xf, (kv.first >> 4) & 0xf, kv.first & 0xf);
        for (int b : kv.second) {
            const unsigned long long t = ((unsigned long long)h[b * 4 + 3] << 32) | h[b * 4 + 2];
            printf("  block %3d tg %u simd %u wave %u t+%llu", b, (h[b * 4] >> 16) & 0xf, (h[b * 4] >> 4) & 3, h[b * 4] & 0xf, t - t0);
        }
        printf("\n");
    }
    return 0;
}
