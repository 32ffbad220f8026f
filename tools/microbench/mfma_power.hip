// What the MFMA pipe delivers at the chip's power limit, by operand data: a chip-filling grid of waves that do
// nothing but v_mfma_f32_32x32x16_f16 (or 16x16x32) on register operands -- zeros, uniform random [-1, 1) or
// SiLU-like values.  TFLOP/s from wall time; run under `rocprofv3 --pmc GRBM_GUI_ACTIVE` for the clock.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_kernel(const half8* __restrict__ src, float* out, int iters) {
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(threadIdx.x + 256 * i) & 1023];
        b[i] = src[(threadIdx.x + 256 * i + 77) & 1023];
    }
    float s = 0;
    if (SHAPE == 32) {
        floatx16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        floatx4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3], b[i & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<_Float16> h(1024 * 8);
    half8* src;
    float* out;
    hipMalloc(&src, h.size() * 2);
    hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
    unsigned seed = 1;
    const auto rnd = [&] {
        seed = seed * 1664525u + 1013904223u;
        return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f;
    };
    const char* names[] = {"zeros", "uniform [-1,1)", "SiLU-like", "small integers"};
    for (int mode = 0; mode < 4; ++mode) {
        for (auto& v : h) {
            float x = rnd();
            if (mode == 0) x = 0;
            if (mode == 2) {
                const float g = rnd() + rnd() + x;
                x = g / (1.f + std::exp(-g));
            }
            if (mode == 3) x = (float)(int)(x * 4);
            v = (_Float16)x;
        }
        hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int shape : {32, 16})
            for (int wps : {1, 2}) {
                const int blocks = cus * wps;
                const int iters = 20000;
                hipEvent_t e0, e1;
                hipEventCreate(&e0), hipEventCreate(&e1);
                auto launch = [&](int it) {
                    if (shape == 32)
                        mfma_kernel<32><<<blocks, 256>>>(src, out, it);
                    else
                        mfma_kernel<16><<<blocks, 256>>>(src, out, it);
                };
                launch(2000);
                hipEventRecord(e0);
                launch(iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double n_mfma = (double)blocks * 4 * iters * (shape == 32 ? 16 : 32);
                const double flop = n_mfma * (shape == 32 ? 32768.0 : 16384.0);
                printf("%-16s %dx%d  %d wave(s)/SIMD: %8.1f TFLOP/s  (%.2f ms)\n", names[mode], shape, shape, wps, flop / ms / 1e9, ms);
            }
    }
    return 0;
}
