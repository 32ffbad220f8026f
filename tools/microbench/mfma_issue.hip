// How fast ONE wave issues independent v_mfma_f32_32x32x16_f16 (6 accumulators, each reused every 6th instruction, as a
// conv_t32 matrix segment does), against 2 and 4 waves per SIMD: cycles per MFMA per SIMD at the nominal clock, from wall
// time, on random operands.  build + run: hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int AGPR>
__global__ __launch_bounds__(256) void k(const half8* __restrict__ src, float* out, int iters) {
    half8 a[2], b[3];
    for (int i = 0; i < 2; ++i) a[i] = src[(threadIdx.x + 256 * i) & 1023];
    for (int i = 0; i < 3; ++i) b[i] = src[(threadIdx.x + 256 * i + 77) & 1023];
    floatx16 acc[6];
    for (int i = 0; i < 6; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (AGPR)
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(b[i % 3]), "v"(a[i / 3]));
                else
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b[i % 3]), "v"(a[i / 3]));
            }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0;
    for (int i = 0; i < 6; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<_Float16> h(1024 * 8);
    unsigned seed = 1;
    for (auto& v : h) {
        seed = seed * 1664525u + 1013904223u;
        v = (_Float16)(((seed >> 8) & 0xffff) / 32768.0f - 1.0f);
    }
    half8* src;
    float* out;
    hipMalloc(&src, h.size() * 2);
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int agpr = 0; agpr < 2; ++agpr)
        for (int w : {1, 2, 4}) {
            const int blocks = cus * w, iters = 20000;
            hipEvent_t e0, e1;
            hipEventCreate(&e0), hipEventCreate(&e1);
            if (agpr) k<1><<<blocks, 256>>>(src, out, 100); else k<0><<<blocks, 256>>>(src, out, 100);
            hipEventRecord(e0);
            if (agpr) k<1><<<blocks, 256>>>(src, out, iters); else k<0><<<blocks, 256>>>(src, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double mfma_per_simd = (double)iters * 12 * w;
            const double tflops = mfma_per_simd * cus * 4 * 32768.0 / (ms * 1e-3) / 1e12;
            printf("%s accumulators, %d wave(s) per SIMD: %.1f ns per MFMA per SIMD = %.1f cycles at %.0f MHz nominal, %.0f TFLOP/s\n",
                   agpr ? "AccVGPR" : "ArchVGPR", w, ms * 1e6 / mfma_per_simd, ms * 1e-3 * p.clockRate * 1e3 / mfma_per_simd, p.clockRate / 1e3, tflops);
        }
    return 0;
}
