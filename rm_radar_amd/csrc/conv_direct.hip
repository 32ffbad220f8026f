// conv_direct.hip -- the small-batch (latency) convolution: MFMA fragments straight from L2, the K
// loop split across the waves of a workgroup.
//
// At batch 1..4 a YOLOv8m layer has only M = 400..6400 output pixels.  The tiled kernels
// (conv_igemm / conv_dma / conv_halo) then run one workgroup per CU at most, and their K loop --
// stage a slice in LDS, barrier, 6..24 MFMAs per wave, repeat 27..81 times -- is a serial chain of
// fixed costs (measured 1800-2000 cycles per slice at 64 x 96, of which the MFMAs are 200-600):
// 20-30 us for a layer whose arithmetic is 2 us of the chip.  This kernel removes the chain:
//
//   * no staging and no barrier in the K loop: every wave loads its MFMA operands directly from
//     global memory in fragment shape (16 B per lane: 8 consecutive k of one pixel / one output
//     channel), three K steps in flight in registers.  Padding taps and rows past M are zero-filled
//     by the buffer bounds check (masked lanes get an out-of-range offset);
//   * each wave computes the WHOLE BM x BN tile for its own contiguous share of the K steps
//     (intra-workgroup split-K), so all 4 or 8 waves of a small tile pull on L2 at once and
//     a weight fragment is reused MREP times from registers;
//   * one barrier at the end: every wave parks its partial tile in LDS, then wave w sums the tiles
//     t = w (mod waves) over the waves IN WAVE ORDER (deterministic) and runs the fused epilogue.
//
// It needs Cin % 32 == 0 (a K step is 32 channels of one tap) and reads operands uncoalesced in
// 64-byte pieces, so it loses to the staged kernels once M is large enough to fill the chip; the
// per-(layer, batch) autotuner decides.
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include "conv_igemm.h"

#ifndef RMR_CONV_TIMING_BUILD
#define RMR_CONV_TIMING_BUILD 0
#endif

namespace rmr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

// v * rcp(1 + e^-v): the hardware reciprocal (1 ulp) instead of an IEEE division -- the epilogue's VALU
// work is not small beside a short K loop (48 values per lane per tile)
__device__ __forceinline__ float silu_d(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

template <int MREP, int NREP, int NW>
__global__ __launch_bounds__(NW * 64) void conv_direct_kernel(const ConvArgs a) {
    constexpr int BM = MREP * 16, BN = NREP * 16;
    constexpr int TILES = MREP * NREP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if RMR_CONV_TIMING_BUILD
    const bool timed = a.timing != nullptr && blockIdx.x == gridDim.x / 2 && wave == 1;
#else
    constexpr bool timed = false;  // stamps are compiled in only with -DRMR_CONV_TIMING_BUILD=1
#endif
    long long tt[6] = {0, 0, 0, 0, 0, 0};
    if (timed) tt[0] = __builtin_readcyclecounter();
    const int px = lane & 15;
    const int kg = lane >> 4;

    const int nt_count = a.Cout_pad / BN;
    const int m0 = (blockIdx.x / nt_count) * BM;
    const int n0 = (blockIdx.x % nt_count) * BN;

    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const auto wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
    constexpr unsigned OOB = 0xffffffffu;

    // ---- per-lane operand bookkeeping ---------------------------------------------------------
    // A fragment i: pixel m0 + 16 i + px; byte offset of its (kh = 0, kw = 0) tap, valid-tap bits
    unsigned pbase[MREP], amask[MREP];
    const int taps = a.KH * a.KW;
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = m0 + i * 16 + px;
        unsigned mask = 0;
        int base = 0;
        if (m < a.M) {
            const int hw = a.Ho * a.Wo;
            const int n = m / hw, rem = m - n * hw;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            base = (((n * a.H + iy0) * a.W + ix0) * a.in_cs + a.in_co + kg * 8) * 2;
            for (int t = 0; t < taps; ++t) {
                const int iy = iy0 + t / a.KW, ix = ix0 + t % a.KW;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) mask |= 1u << t;
            }
        }
        pbase[i] = (unsigned)base;
        amask[i] = mask;
    }
    // B fragment j: output channel n0 + 16 j + px (the K offset rides in the scalar offset)
    unsigned wbase[NREP];
#pragma unroll
    for (int j = 0; j < NREP; ++j) wbase[j] = (unsigned)(((n0 + j * 16 + px) * a.Kp + kg * 8) * 2);

    // ---- this wave's share of the K steps (a step = 32 channels of one tap) ---------------------
    const int chunks = a.Cin / 32;
    const int steps = taps * chunks;
    const int s_begin = wave * steps / NW, s_end = (wave + 1) * steps / NW;
    const int n_own = s_end - s_begin;

    half8 xf[3][MREP], wf[3][NREP];
    auto load = [&](int st, int buf) {
        st = st < s_end ? st : s_end - 1;  // the tail re-fetches the last step instead of branching
        st = st < 0 ? 0 : st;
        const int tap = st / chunks, cc = st - tap * chunks;
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const unsigned a_off = (unsigned)(((kh * a.W + kw) * a.in_cs + cc * 32) * 2);
        const int b_off = (tap * a.Cin + cc * 32) * 2;
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            const unsigned off = ((amask[i] >> tap) & 1u) ? pbase[i] + a_off : OOB;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
            xf[buf][i] = __builtin_bit_cast(half8, v);
        }
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wt_rsrc, wbase[j], b_off, 0);
            wf[buf][j] = __builtin_bit_cast(half8, v);
        }
    };

    floatx4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int buf) {
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[buf][j], xf[buf][i], acc[i][j], 0, 0, 0);
    };

    if (timed) tt[1] = __builtin_readcyclecounter();
    if (n_own > 0) {
        load(s_begin, 0);
        load(s_begin + 1, 1);
        for (int k = 0; k < n_own; k += 3) {
            load(s_begin + k + 2, 2);
            mma(0);
            if (timed && k == 0) {
                asm volatile("s_nop 0" ::"v"(acc[0][0]));
                tt[2] = __builtin_readcyclecounter();
            }
            load(s_begin + k + 3, 0);
            if (k + 1 < n_own) mma(1);
            load(s_begin + k + 4, 1);
            if (k + 2 < n_own) mma(2);
        }
    }

    // ---- park the partial tile, then reduce tile t = wave (mod NW) over the waves in order -----
    if (timed) {
        asm volatile("s_nop 0" ::"v"(acc[0][0]));
        tt[3] = __builtin_readcyclecounter();
    }
    float* const part = (float*)smem;  // [NW][TILES][64 lanes][4]
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j)
            *(floatx4*)(part + ((size_t)(wave * TILES + i * NREP + j) * 64 + lane) * 4) = acc[i][j];
    __syncthreads();
    if (timed) tt[4] = __builtin_readcyclecounter();

    const int cq = kg * 4;
    for (int t = wave; t < TILES; t += NW) {
        floatx4 sum = *(const floatx4*)(part + ((size_t)t * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const floatx4 p = *(const floatx4*)(part + ((size_t)(w * TILES + t) * 64 + lane) * 4);
            sum[0] += p[0];
            sum[1] += p[1];
            sum[2] += p[2];
            sum[3] += p[3];
        }
        const int i = t / NREP, j = t - i * NREP;
        const int m = m0 + i * 16 + px;
        if (m >= a.M) continue;
        const int n = n0 + j * 16 + cq;
        const float4 b = *(const float4*)(a.bias + n);
        float v[4] = {sum[0] + b.x, sum[1] + b.y, sum[2] + b.z, sum[3] + b.w};
        if (a.act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_d(v[e]);
        }
        if (a.res) {
            union {
                uint2 u;
                _Float16 h[4];
            } rr;
            rr.u = *(const uint2*)((const _Float16*)a.res + (long)m * a.res_cs + a.res_co + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rr.h[e];
        }
        if (a.out32) {
            *(float4*)(a.out32 + (long)m * a.out_cs + a.out_co + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            union {
                uint2 u;
                _Float16 h[4];
            } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.h[e] = (_Float16)v[e];
            *(uint2*)((_Float16*)a.out + (long)m * a.out_cs + a.out_co + n) = o.u;
        }
    }
    if (timed && lane == 0) {
        tt[5] = __builtin_readcyclecounter();
        for (int q = 0; q < 5; ++q) a.timing[q] = tt[q + 1] - tt[q];
        a.timing[5] = n_own;
    }
}

struct DirectTile {
    int bm, bn, waves;
    void (*kernel)(const ConvArgs);
};

#define DTILE(MR, NR, NW) \
    { MR * 16, NR * 16, NW, conv_direct_kernel<MR, NR, NW> }

const DirectTile kDirectTiles[] = {
    DTILE(4, 6, 4),  // 0: 64 x 96, K over 4 waves
    DTILE(2, 6, 8),  // 1: 32 x 96, 8 waves
    DTILE(4, 3, 8),  // 2: 64 x 48, 8 waves
    DTILE(2, 3, 8),  // 3: 32 x 48, 8 waves
    DTILE(1, 6, 8),  // 4: 16 x 96, 8 waves
    DTILE(1, 3, 8),  // 5: 16 x 48, 8 waves
    DTILE(2, 6, 4),  // 6: 32 x 96, 4 waves
    DTILE(4, 4, 4),  // 7: 64 x 64, 4 waves
    DTILE(2, 4, 8),  // 8: 32 x 64, 8 waves
    DTILE(1, 4, 8),  // 9: 16 x 64, 8 waves
    DTILE(4, 3, 4),  // 10: 64 x 48, 4 waves
    DTILE(1, 1, 8),  // 11: 16 x 16, 8 waves (class / box heads with 16 output channels)
};
constexpr int kNumDirectTiles = sizeof(kDirectTiles) / sizeof(kDirectTiles[0]);

int direct_lds_bytes(const DirectTile& t) { return t.waves * t.bm * t.bn * 4; }

}  // namespace

int conv_direct_num_tiles() { return kNumDirectTiles; }
ConvTile conv_direct_tile(int id) { return ConvTile{kDirectTiles[id].bm, kDirectTiles[id].bn, 32}; }

bool conv_direct_supported(const ConvArgs& a, int tile) {
    if (a.Cin % 32 || a.KH * a.KW > 16 || a.in_bytes == 0 || a.wt_bytes == 0 || a.pre) return false;
    if (tile < 0) return true;
    if (tile >= kNumDirectTiles) return false;
    const DirectTile& t = kDirectTiles[tile];
    return a.Cout_pad % t.bn == 0 && direct_lds_bytes(t) <= 160 * 1024;
}

void launch_conv_direct(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile) {
    if (tile < 0 || tile >= kNumDirectTiles) fail(RMR_ERR_INVALID_ARGUMENT, "conv_direct: tile %d out of range", tile);
    if (!conv_direct_supported(a, tile)) fail(RMR_ERR_LOGIC, "conv_direct: layer not supported by tile %d", tile);
    const DirectTile& t = kDirectTiles[tile];
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 4 || a.out_co % 4) fail(RMR_ERR_LOGIC, "conv_direct: misaligned view");
    if (a.in_bytes > 0xf0000000ull) fail(RMR_ERR_LOGIC, "conv_direct: input view larger than 3.75 GiB");
    static std::once_flag once;
    std::call_once(once, [] {
        for (const DirectTile& d : kDirectTiles)
            (void)hipFuncSetAttribute((const void*)d.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int grid = ((a.M + t.bm - 1) / t.bm) * (a.Cout_pad / t.bn);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d x%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, tile);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    t.kernel<<<grid, t.waves * 64, direct_lds_bytes(t), stream>>>(a);
    RMR_HIP(hipGetLastError());
}

}  // namespace rmr
