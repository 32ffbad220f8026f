// What clock do short kernels run at?  A small grid runs a chain of dependent VALU adds and reads both counters (s_memtime: shader
// clock, s_memrealtime: 100 MHz) around it; launched back to back (as the layers of a batch-1 frame are) and after idle pauses.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/microbench/clock_probe.hip -o /tmp/clk && /tmp/clk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
#include <vector>
#include <algorithm>

__global__ void probe(unsigned long long* out, int n, float seed) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = seed + threadIdx.x;
    for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f;   // dependent chain: 2 VALU per iteration with -ffp-contract=off, 1 fma otherwise
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = c1 - c0, out[1] = r1 - r0;
    if (x == 12345.678f) out[2] = 1;
}

int main() {
    unsigned long long* d;
    (void)hipMalloc(&d, 64);
    unsigned long long h[2];
    for (int grid : {1, 256, 1024}) {
        for (int n : {500, 4000, 40000}) {
            for (int mode = 0; mode < 2; ++mode) {   // 0: back to back, 1: 2 ms of idle before each launch
                std::vector<double> mhz, cyc;
                for (int rep = 0; rep < 30; ++rep) {
                    if (mode) usleep(2000);
                    else for (int k = 0; k < 20; ++k) probe<<<grid, 256>>>(d, n, 1.f);
                    probe<<<grid, 256>>>(d, n, 1.f);
                    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                    mhz.push_back((double)h[0] / ((double)h[1] * 0.01));
                    cyc.push_back((double)h[0] / n);
                }
                std::sort(mhz.begin(), mhz.end());
                std::sort(cyc.begin(), cyc.end());
                printf("grid %4d  chain %6d  %-12s: shader clock p50 %7.1f MHz (min %7.1f max %7.1f), %.2f shader cycles per iteration\n", grid, n,
                       mode ? "after idle" : "back to back", mhz[15], mhz[0], mhz[29], cyc[15]);
            }
        }
    }
    return 0;
}
