// Issue rates of v_fma_f32 and the transcendentals on a chip-filling grid (gfx950): cycles per wave
// instruction at the nominal clock, from wall time.  Measured: fma 2.6 (SIMD-32: 2 at the peak), v_exp_f32 /
// v_rcp_f32 10.4 at four waves per SIMD -- the transcendentals run at a quarter of the FMA rate, which is
// why a SiLU (one of each) costs as much as eight plain VALU instructions in the short-K epilogues.
// (An MFMA loop was dropped from this file: hipcc rotated the accumulator tuples of the unrolled body so
// that srcC overlapped other MFMAs' destinations, and the number measured the stalls that causes.)
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 0.001f * (threadIdx.x + i + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] * 1.0001f + 0.5f;                          // v_fma_f32
            if (OP == 1) a[i] = __builtin_amdgcn_exp2f(a[i]);                   // v_exp_f32
            if (OP == 2) a[i] = __builtin_amdgcn_rcpf(a[i]);                    // v_rcp_f32
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
double run(const char* name, int waves_per_simd) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    rate_kernel<OP><<<blocks, 256>>>(out, 100);
    hipEventRecord(e0);
    rate_kernel<OP><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double clock_hz = p.clockRate * 1e3;
    const double insts_per_simd = (double)iters * 8 * waves_per_simd;
    const double cyc = ms * 1e-3 * clock_hz / insts_per_simd;
    printf("%-28s waves/SIMD %d: %.2f cycles per wave instruction (%.3f ms, clock %.0f MHz)\n", name, waves_per_simd, cyc, ms,
           clock_hz / 1e6);
    hipFree(out);
    return cyc;
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w);
        run<1>("v_exp_f32", w);
        run<2>("v_rcp_f32", w);
    }
    return 0;
}
