// scalar_atomic.hip -- does s_atomic_add (scalar memory atomic with return, lgkmcnt) work on gfx950, and what does one cost?
// Every workgroup draws `draws` tickets from one counter with its first wave; the host checks that the tickets are a
// permutation of 0 .. n-1 and prints the average latency of a draw (s_memtime around it).
// build: hipcc --offload-arch=gfx950 -O3 -o scalar_atomic scalar_atomic.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void draw_scalar(unsigned* ctr, unsigned* out, unsigned long long* cycles, int draws) {
    if (threadIdx.x >= 64) return;
    unsigned long long t = 0;
    for (int i = 0; i < draws; ++i) {
        unsigned one = 1, got;
        const unsigned long long t0 = __builtin_readcyclecounter();
        asm volatile("s_mov_b32 %0, %2\n\ts_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(got) : "s"(ctr), "s"(one) : "memory");
        t += __builtin_readcyclecounter() - t0;
        if (threadIdx.x == 0) out[(blockIdx.x * draws) + i] = got;
    }
    if (threadIdx.x == 0) cycles[blockIdx.x] = t;
}

__global__ void draw_vector(unsigned* ctr, unsigned* out, unsigned long long* cycles, int draws) {
    if (threadIdx.x >= 64) return;
    unsigned long long t = 0;
    for (int i = 0; i < draws; ++i) {
        unsigned got = 0;
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) got = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        got = __builtin_amdgcn_readfirstlane(got);
        t += __builtin_readcyclecounter() - t0;
        if (threadIdx.x == 0) out[(blockIdx.x * draws) + i] = got;
    }
    if (threadIdx.x == 0) cycles[blockIdx.x] = t;
}

int run(int grid, int draws);
int main() {
    // 512 workgroups at once on one counter; then ONE workgroup (the latency of an uncontended draw)
    return run(512, 8) || run(1, 64) || run(8, 64);
}

int run(const int grid, const int draws) {
    const int n = grid * draws;
    unsigned *ctr, *out;
    unsigned long long* cyc;
    hipMalloc(&ctr, 4);
    hipMalloc(&out, n * 4);
    hipMalloc(&cyc, grid * 8);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(ctr, 0, 4);
        hipMemset(out, 0xff, n * 4);
        if (mode == 0)
            draw_scalar<<<grid, 256>>>(ctr, out, cyc, draws);
        else
            draw_vector<<<grid, 256>>>(ctr, out, cyc, draws);
        const hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            std::printf("%s: %s\n", mode ? "vector" : "scalar", hipGetErrorString(e));
            return 1;
        }
        std::vector<unsigned> h(n);
        std::vector<unsigned long long> c(grid);
        hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        bool perm = true;
        for (int i = 0; i < n; ++i) perm = perm && h[i] == (unsigned)i;
        double avg = 0;
        for (auto v : c) avg += (double)v / draws;
        std::printf("%s atomic: tickets are %sa permutation of 0..%d, %.0f cycles per draw (%d workgroups drawing at once)\n",
                    mode ? "vector" : "scalar", perm ? "" : "NOT ", n - 1, avg / grid, grid);
    }
    hipFree(ctr);
    hipFree(out);
    hipFree(cyc);
    return 0;
}
