// conv_ws_s2.hip -- the second layer of YOLOv8m: 3x3 / stride 2 / pad 1, 48 -> 96 channels,
// 320x320 -> 160x160, weights stationary in registers (the stride-2 sibling of conv_ws.hip).
//
// Like the first layer it is an HBM stream (96 B in per input pixel, 48 B out: 3.8 GB per 256
// images) that the generic im2col kernel runs at 2.2 TB/s.  Here:
//   * a 4-wave workgroup owns a strip of output rows of ONE HALF of the map (80 output columns);
//     wave (r, h) computes output row 2 s + r of step s for channels 48 h .. 48 h + 47 and keeps
//     that half of the filter (48 x 432) in 168 VGPRs for the whole strip;
//   * input rows are DMA'd once into a 9-slot LDS ring, four new rows per step, one step ahead.
//     A ring row holds the 162 input pixels the 80 outputs touch, DE-INTERLEAVED by column parity
//     (plane 0: columns 2 x0 - 1 + 2 k, plane 1: columns 2 x0 + 2 k): the three taps of output
//     column o are plane0[o], plane1[o], plane0[o + 1], so the 16 lanes of a fragment read
//     consecutive 96-byte pixels -- the conflict-free pattern of conv_ws -- instead of every other
//     one.  Columns / rows outside the image arrive as zeros from the buffer bounds check;
//   * results are packed to f16 in registers and stored during the NEXT step.  Neither the 16 row DMAs nor the 15
//     stores of a step are issued in a block any more (a vector-memory instruction costs its wave 60-180 issue
//     cycles, and with one wave per SIMD nothing else runs meanwhile): the DMAs ride behind the MFMAs of K-steps
//     0..3, the stores behind those of K-steps 4..13, bounds-checked buffer stores that are always issued, so the
//     wait that admits a step's rows is a COUNTED one -- the 15 stores issued after those DMAs may still be in
//     flight (loads and stores share vmcnt and retire in order).
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include "conv_igemm.h"

namespace rmr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int S2_C = 48;                 // input channels
constexpr int S2_WO = 160;               // output width (input 320)
constexpr int S2_PIX = S2_C * 2;         // 96 bytes per input pixel
constexpr int S2_PLANE = 81 * S2_PIX;    // 81 pixels per parity plane
constexpr int S2_SLOT = 16384;           // 2 planes (15552 B) padded to 16 DMA KiB
constexpr int S2_SLOTS = 9;
constexpr int S2_KSTEPS = 14;            // ceil(9 * 48 / 32)
constexpr int S2_LDS = S2_SLOTS * S2_SLOT;

__device__ __forceinline__ float silu_2(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ void dma16_2(u32x4 rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc)
                 : "memory");
}

// strip_rows: output rows per workgroup (even, divides Ho)
template <bool ACT, bool OUT32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv_ws_s2_kernel(const ConvArgs a, const int strip_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = wave >> 1;   // output row of the step
    const int nh = wave & 1;   // channel half
    const int frow = lane & 15;
    const int kg = lane >> 4;
    const bool hi = kg >= 2;

    const int strips = a.Ho / strip_rows;
    int b = blockIdx.x;
    const int xhalf = b & 1;
    b >>= 1;
    const int img = b / strips;
    const int y_base = (b % strips) * strip_rows;
    const int steps = strip_rows / 2;
    const int x0 = xhalf * 80;        // first output column
    const int ix0 = 2 * x0 - 1;       // input column of ring entry e = 0
    const int iy_base = 2 * y_base - 1;  // input row of relative row ry = 0

    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu),
                           sgpr(a.in_bytes), sgpr(0x00020000u)};

    // ---- this wave's half of the filter, as B fragments ---------------------------------------
    half8 wreg[S2_KSTEPS][3];
#pragma unroll
    for (int ks = 0; ks < S2_KSTEPS; ++ks)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            wreg[ks][j] = *(const half8*)((const _Float16*)a.wt + (size_t)(nh * 48 + j * 16 + frow) * a.Kp + ks * 32 + kg * 8);
#pragma unroll
    for (int ks = 0; ks < S2_KSTEPS; ++ks)
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(wreg[ks][j]));  // keep them in registers (see conv_ws.hip)

    // ---- DMA: ring row = 16 instructions; instruction i of a row, lane l -> chunk g = 64 i + l ----
    // wave w issues instructions w, w + 4, w + 8, w + 12 of every row
    unsigned goff[4], gbad[4];  // byte offset of the chunk within its image row; all ones where there is none
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = (wave + 4 * j) * 64 + lane;
        const int plane = g / 486, k = (g % 486) / 6, ch = g % 6;
        const int ix = ix0 + 2 * k + plane;
        const bool ok = g < 972 && ix >= 0 && ix < a.W;  // padding columns and the tail of the 16th KiB
        goff[j] = ok ? (unsigned)((ix * a.in_cs + a.in_co + ch * 8) * 2) : 0u;
        gbad[j] = ok ? 0u : 0xffffffffu;
    }
    const unsigned row_bytes = (unsigned)(a.W * a.in_cs * 2);
    auto issue_row = [&](int ry, bool live = true) {
        const int iy = iy_base + ry;
        const unsigned dead = (live && iy >= 0 && iy < a.H) ? 0u : 0xffffffffu;
        const unsigned rowoff = (unsigned)(img * a.H + iy) * row_bytes;
        const unsigned slot = lds0 + (unsigned)(ry % S2_SLOTS) * S2_SLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16_2(in_rsrc, sgpr(slot + (wave + 4 * j) * 1024), (goff[j] + rowoff) | dead | gbad[j]);
    };
#pragma unroll 1
    for (int ry = 0; ry < 5; ++ry) issue_row(ry);  // rows of step 0

    const int px = lane & 15;
    const int cq = (lane >> 4) * 4;
    float4 bias[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) bias[j] = *(const float4*)(a.bias + nh * 48 + j * 16 + cq);

    const unsigned lane_off = (unsigned)(frow * S2_PIX + (kg & 1) * 16);
    u32x2 outv[5][3];
    long m_row = 0;
    // f16 results leave through a bounds-checked resource: a cell of the step "before the first one" is stored at an
    // out-of-range offset (dropped), so the store count per step is a constant
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0xfffffff0u, 0x00020000);
    unsigned out_row = 0xffffffffu;   // byte offset of (m_row of the PREVIOUS step, this lane's first channel); all ones: none
    const unsigned out_lane = (unsigned)(px * a.out_cs + a.out_co + nh * 48 + cq) * 2u;
    auto store_cell = [&](int c) {   // c: compile-time at every call site
        if (OUT32) return;
        const int i = c / 3, j = c % 3;
        const unsigned off = out_row == 0xffffffffu ? 0xffffffffu : out_row + out_lane + (unsigned)(i * 16 * a.out_cs + j * 16) * 2u;
        __builtin_amdgcn_raw_buffer_store_b64(outv[i][j], out_rsrc, off, 0, 0);
    };

    unsigned vb[3];
    auto read_frags = [&](int ks, half8* xf) {
        // k = 32 ks + 8 kg: lanes 0-31 start at kL, lanes 32-63 at kL + 16 (same or next tap); tap (kh, kw)
        // of output column o reads plane 0 / 1 / 0 at pixel o / o / o + 1
        const int kL = 32 * ks, kH = 32 * ks + 16;
        const int tapL = kL / S2_C, cL = kL % S2_C;
        const int tapH = kH < 9 * S2_C ? kH / S2_C : tapL, cH = kH < 9 * S2_C ? kH % S2_C : cL;
        const int kwL = tapL % 3, kwH = tapH % 3;
        const unsigned immL = (unsigned)((kwL == 1 ? S2_PLANE : kwL == 2 ? S2_PIX : 0) + cL * 2);
        const unsigned immH = (unsigned)((kwH == 1 ? S2_PLANE : kwH == 2 ? S2_PIX : 0) + cH * 2);
        const unsigned addr = hi ? vb[tapH / 3] + immH : vb[tapL / 3] + immL;
        const __attribute__((address_space(3))) unsigned char* p =
            (const __attribute__((address_space(3))) unsigned char*)(size_t)addr;
#pragma unroll
        for (int i = 0; i < 5; ++i) xf[i] = *(const __attribute__((address_space(3))) half8*)(p + i * 16 * S2_PIX);
    };

    for (int s = 0; s < steps; ++s) {
        // this step's rows were issued during the previous step, BEFORE that step's 15 stores (none in step 0, none
        // with an f32 output): everything older than those stores has landed
        if (OUT32 || s <= 1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // ... for every wave; step s - 1 is fully consumed
        const bool more = s + 1 < steps;
        out_row = s > 0 ? (unsigned)(m_row * a.out_cs) * 2u : 0xffffffffu;
        const int y = y_base + 2 * s + r;
        m_row = ((long)img * a.Ho + y) * S2_WO + x0;

#pragma unroll
        for (int kh = 0; kh < 3; ++kh) vb[kh] = lds0 + ((4 * s + 2 * r + kh) % S2_SLOTS) * S2_SLOT + lane_off;

        floatx4 acc[5][3];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

        half8 xf[2][5];
        read_frags(0, xf[0]);
#pragma unroll
        for (int ks = 0; ks < S2_KSTEPS; ++ks) {
            if (ks + 1 < S2_KSTEPS) read_frags(ks + 1, xf[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 8; ++c)
                acc[c / 3][c % 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[ks][c % 3], xf[ks & 1][c / 3], acc[c / 3][c % 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the step's vector-memory work, one piece behind the MFMAs of every K-step: the four rows of step s + 1
            // (out of range behind the strip's last step: the counts stay constant), then the 15 cells of step s - 1
            if (ks < 4) {
                issue_row(4 * s + 5 + ks, more);
            } else if (ks < 9) {
                store_cell(2 * (ks - 4));
                store_cell(2 * (ks - 4) + 1);
            } else {
                store_cell(10 + (ks - 9));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 8; c < 15; ++c)
                acc[c / 3][c % 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[ks][c % 3], xf[ks & 1][c / 3], acc[c / 3][c % 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float4 bb = bias[j];
                float v[4] = {acc[i][j][0] + bb.x, acc[i][j][1] + bb.y, acc[i][j][2] + bb.z, acc[i][j][3] + bb.w};
                if (ACT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_2(v[e]);
                }
                if (OUT32) {
                    *(float4*)(a.out32 + (m_row + i * 16 + px) * a.out_cs + a.out_co + nh * 48 + j * 16 + cq) =
                        make_float4(v[0], v[1], v[2], v[3]);
                    continue;
                }
                union {
                    u32x2 u;
                    _Float16 h[4];
                } o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o.h[e] = (_Float16)v[e];
                outv[i][j] = o.u;
            }
        }
    }
    out_row = (unsigned)(m_row * a.out_cs) * 2u;   // the strip's last step
#pragma unroll
    for (int c = 0; c < 15; ++c) store_cell(c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

const int kS2StripRows[] = {40, 20, 10, 8, 4, 2};
constexpr int kNumS2 = sizeof(kS2StripRows) / sizeof(kS2StripRows[0]);

}  // namespace

int conv_ws_s2_num_variants() { return kNumS2; }

bool conv_ws_s2_supported(const ConvArgs& a, int variant) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 2 || a.pad != 1 || a.res) return false;
    if (a.Cin != S2_C || a.Cout_pad != 96 || a.Wo != S2_WO || a.W != 2 * S2_WO || a.H != 2 * a.Ho) return false;
    if (a.Kp < S2_KSTEPS * 32 || (!a.out32 && !a.out) || a.in_bytes == 0) return false;
    if (variant < 0) return a.Ho % 2 == 0;
    return variant < kNumS2 && a.Ho % kS2StripRows[variant] == 0;
}

void launch_conv_ws_s2(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int variant) {
    if (variant < 0 || variant >= kNumS2) fail(RMR_ERR_INVALID_ARGUMENT, "conv_ws_s2: variant %d out of range", variant);
    if (!conv_ws_s2_supported(a, variant)) fail(RMR_ERR_LOGIC, "conv_ws_s2: layer not supported by variant %d", variant);
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 4 || a.out_co % 4) fail(RMR_ERR_LOGIC, "conv_ws_s2: misaligned view");
    if (a.in_bytes > 0xf0000000ull) fail(RMR_ERR_LOGIC, "conv_ws_s2: input view larger than 3.75 GiB");
    if (!a.out32 && (double)a.M * a.out_cs * 2 >= 4.0e9) fail(RMR_ERR_LOGIC, "conv_ws_s2: output view of 4 GB or more (32-bit store offsets)");
    using Kern = void (*)(const ConvArgs, int);
    static const Kern kernels[4] = {conv_ws_s2_kernel<false, false>, conv_ws_s2_kernel<false, true>,
                                    conv_ws_s2_kernel<true, false>, conv_ws_s2_kernel<true, true>};
    static std::once_flag once;
    std::call_once(once, [] {
        for (Kern k : kernels)
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int sr = kS2StripRows[variant];
    const int grid = a.N * 2 * (a.Ho / sr);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d v%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, variant);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    kernels[(a.act ? 2 : 0) + (a.out32 ? 1 : 0)]<<<grid, 256, S2_LDS, stream>>>(a, sr);
    RMR_HIP(hipGetLastError());
}

}  // namespace rmr
