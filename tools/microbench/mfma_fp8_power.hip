// The MFMA pipe at the power limit for the fp8 forms (companion of mfma_power.hip): non-scaled
// v_mfma_f32_32x32x16_fp8_fp8 (f16 rate, half the operand bytes) and the MX form
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (twice the f16 rate), on random e4m3 operands.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_fp8_power.hip -o /tmp/mfma_fp8 && /tmp/mfma_fp8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx8 __attribute__((ext_vector_type(8)));

template <int FORM>
__global__ __launch_bounds__(256) void mfma_kernel(const int* __restrict__ src, float* out, int iters) {
    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float s = 0;
    if (FORM == 0) {
        long a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = ((const long*)src)[(threadIdx.x + 256 * i) & 1023];
            b[i] = ((const long*)src)[(threadIdx.x + 256 * i + 77) & 1023];
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
        }
    } else {
        intx8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                a[i][r] = src[(threadIdx.x * 8 + r + 2048 * i) & 8191];
                b[i][r] = src[(threadIdx.x * 8 + r + 2048 * i + 777) & 8191];
            }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    // cbsz = 0 / blgp = 0: both operands fp8 (e4m3); scales: E8M0 127 = 1.0 in every byte
                    acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(i + u) & 3], b[i], acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<unsigned char> h(8192 * 4);
    int* src;
    float* out;
    hipMalloc(&src, h.size());
    hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
    unsigned seed = 1;
    const auto rnd = [&] {
        seed = seed * 1664525u + 1013904223u;
        return seed >> 8;
    };
    const char* names[] = {"zeros", "random e4m3 |x| < 2"};
    for (int mode = 0; mode < 2; ++mode) {
        for (auto& v : h) {
            // e4m3: sign(1) exp(4, bias 7) mantissa(3): exponents 0..7 -> |x| < 2
            const unsigned r = rnd();
            v = mode == 0 ? 0 : (unsigned char)(((r & 1) << 7) | (((r >> 1) % 8) << 3) | ((r >> 4) & 7));
        }
        hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice);
        for (int form : {0, 1})
            for (int wps : {1, 2}) {
                const int blocks = cus * wps, iters = 20000;
                hipEvent_t e0, e1;
                hipEventCreate(&e0), hipEventCreate(&e1);
                auto launch = [&](int it) {
                    if (form == 0)
                        mfma_kernel<0><<<blocks, 256>>>(src, out, it);
                    else
                        mfma_kernel<1><<<blocks, 256>>>(src, out, it);
                };
                launch(2000);
                hipEventRecord(e0);
                launch(iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double flop = (double)blocks * 4 * iters * 16 * (form == 0 ? 32768.0 : 131072.0);
                printf("%-22s %s  %d wave(s)/SIMD: %8.1f TFLOP/s  (%.2f ms)\n", names[mode],
                       form == 0 ? "32x32x16 fp8 (non-scaled)" : "32x32x64 f8f6f4 (MX, unit scales)", wps, flop / ms / 1e9, ms);
            }
    }
    return 0;
}
