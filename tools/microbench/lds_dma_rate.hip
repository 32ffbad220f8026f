// How fast one CU pulls L2-resident data into LDS with buffer_load_dwordx4 ... lds, by access shape:
//   contiguous KiB per instruction (what conv_t32's pre-packed weights are), or 16 rows of 64 B at a pitch of
//   384 B / 3456 B (activation rows of a 192-channel tensor, weight rows of a [Cout][K] matrix).
// Every workgroup streams its own 64 KiB window (L2-resident after the first pass) `iters` times into a 32 KiB
// ring, 8 instructions in flight per wave.  Prints bytes per clock per CU at the nominal 2.4 GHz and GB/s.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_dma_rate.hip -o /tmp/lds_dma && /tmp/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

template <int PITCH>   // 0: contiguous KiB; else 16 rows of 64 B, PITCH bytes apart
__global__ __launch_bounds__(512) void dma_kernel(const char* src, int iters, int window) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const unsigned lds0 = sgpr((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const char* base = src + (size_t)blockIdx.x * window;
    const u32x4 rsrc = {sgpr((unsigned)(size_t)base), sgpr((unsigned)((size_t)base >> 32) & 0xffffu), sgpr((unsigned)window), sgpr(0x00020000u)};
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned voff = PITCH ? (unsigned)((lane >> 2) * PITCH + (lane & 3) * 16) : (unsigned)lane * 16u;
    const int blocks = PITCH ? window / (16 * PITCH) * (PITCH / 64) : window / 1024;   // KiB-sized pieces in the window
    int b = wave;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned soff;
            if (PITCH) {
                const int grp = b / (PITCH / 64), col = b % (PITCH / 64);   // 16-row group, 64-byte column
                soff = (unsigned)(grp * 16 * PITCH + col * 64);
            } else {
                soff = (unsigned)b * 1024u;
            }
            dma16(rsrc, sgpr(lds0 + ((wave * 8 + k) & 31) * 1024), voff, sgpr(soff));
            b += 8;
            if (b >= blocks) b -= blocks;
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int PITCH>
void run(const char* name, const char* src, int cus, int wg_per_cu) {
    const int window = 64 * 1024, iters = 4000;   // 32 workgroups per XCD x 64 KiB = 2 MiB: stays in the 4 MiB L2
    const int grid = cus * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    dma_kernel<PITCH><<<grid, 512, 32 * 1024>>>(src, 100, window);
    hipEventRecord(e0);
    dma_kernel<PITCH><<<grid, 512, 32 * 1024>>>(src, iters, window);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 8 * 8 * 1024.0 * iters;
    printf("%-44s %d WG/CU: %7.1f GB/s per CU, %5.1f B/clk/CU at 2.4 GHz, %6.2f TB/s chip\n", name, wg_per_cu, bytes / ms / 1e6 / cus,
           bytes / (ms * 1e-3) / cus / 2.4e9, bytes / ms / 1e9);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    char* src;
    hipMalloc(&src, (size_t)cus * 2 * 256 * 1024);
    hipMemset(src, 1, (size_t)cus * 2 * 256 * 1024);
    for (int w : {1, 2}) {
        run<0>("contiguous KiB per instruction", src, cus, w);
        run<384>("16 rows x 64 B, pitch 384 B (192-ch rows)", src, cus, w);
        run<3456>("16 rows x 64 B, pitch 3456 B ([Cout][K] rows)", src, cus, w);
    }
    return 0;
}
