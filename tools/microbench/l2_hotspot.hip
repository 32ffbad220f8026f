// Does the L2 -> LDS path care WHICH addresses the workgroups of an XCD read at the same time?
// Every workgroup streams a window of `window` bytes into LDS with buffer_load_dwordx4 ... lds, contiguous KiB per instruction,
// 8 instructions in flight per wave, `iters` passes.  Modes:
//   own     : every workgroup its own window (tools/microbench/lds_dma_rate.hip: 61 B/clk/CU)
//   shared  : ALL workgroups the SAME window, from its start, in the same order (what the tiles of a convolution layer do
//             with the layer's weights: the same (chunk, tap) slices at about the same time)
//   rotated : the same window, workgroup b starting at block (b * step) % blocks (same bytes, de-correlated order)
// build + run: hipcc --offload-arch=gfx950 -O3 tools/microbench/l2_hotspot.hip -o /tmp/l2hs && /tmp/l2hs
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

__global__ __launch_bounds__(256) void dma_kernel(const char* src, int iters, int window, int mode, int step) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const unsigned lds0 = sgpr((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const char* base = src + (mode == 0 ? (size_t)blockIdx.x * window : 0);
    const u32x4 rsrc = {sgpr((unsigned)(size_t)base), sgpr((unsigned)((size_t)base >> 32) & 0xffffu), sgpr((unsigned)window), sgpr(0x00020000u)};
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned voff = (unsigned)lane * 16u;
    const int blocks = window / 1024;
    int b = wave + (mode == 2 ? (int)((blockIdx.x * (unsigned)step) % (unsigned)blocks) : 0);
    if (b >= blocks) b -= blocks;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            dma16(rsrc, sgpr(lds0 + ((wave * 8 + k) & 31) * 1024), voff, sgpr((unsigned)b * 1024u));
            b += 4;
            if (b >= blocks) b -= blocks;
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

void run(const char* name, const char* src, int cus, int wg_per_cu, int window, int mode, int step) {
    const int iters = 2000;
    const int grid = cus * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    dma_kernel<<<grid, 256, 32 * 1024>>>(src, 100, window, mode, step);
    hipEventRecord(e0);
    dma_kernel<<<grid, 256, 32 * 1024>>>(src, iters, window, mode, step);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 4 * 8 * 1024.0 * iters;
    printf("%-34s window %4d KiB  %d WG/CU: %6.1f B/clk/CU at 2.4 GHz, %6.2f TB/s chip\n", name, window / 1024, wg_per_cu,
           bytes / (ms * 1e-3) / cus / 2.4e9, bytes / ms / 1e9);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    char* src;
    hipMalloc(&src, (size_t)cus * 2 * 1024 * 1024);
    hipMemset(src, 1, (size_t)cus * 2 * 1024 * 1024);
    for (int w : {1, 2}) {
        run("own window", src, cus, w, 64 * 1024, 0, 0);
        for (int window : {64 * 1024, 512 * 1024, 2048 * 1024}) {
            run("shared window, same order", src, cus, w, window, 1, 0);
            run("shared window, rotated by 1 KiB", src, cus, w, window, 2, 1);
            run("shared window, rotated by 4 KiB", src, cus, w, window, 2, 4);
            run("shared window, rotated by 17 KiB", src, cus, w, window, 2, 17);
        }
    }
    return 0;
}
