// conv_igemm.hip -- implicit-GEMM convolution for gfx950: v_mfma_f32_16x16x32_f16, 64-lane
// waves, LDS-staged operand tiles, fused bias + SiLU + residual epilogue.
//
// GEMM view: out[M = N*Ho*Wo][Cout] = im2col(in)[M][K = KH*KW*Cin] * W^T.
// Activations are NHWC f16, so for a fixed (kh,kw) the Cin values of a pixel are contiguous:
// the im2col row is gathered on the fly in 16-byte chunks (8 channels) straight from the
// strided input view, never materialised.  Weights are pre-packed [Cout][K] (K contiguous), i.e.
// both operands are "K-inner" and both fragments are single ds_read_b128 per lane.
//
// The MFMA operands are swapped -- D = W_frag(16 cout x 32 k) * X_frag(32 k x 16 pixels) -- so
// that each lane ends with 4 CONSECUTIVE output channels of one pixel (D row = (lane>>4)*4+r,
// D col = lane&15): the NHWC epilogue is one 8-byte store per 16x16 tile per lane, 32 B
// contiguous per pixel per tile, instead of 2-byte scattered stores.
//
// Work decomposition: 256 threads = 4 waves arranged WM x WN; each wave owns MREP x NREP
// 16x16 accumulator tiles; workgroup tile BM x BN = (WM*MREP*16) x (WN*NREP*16), BK = 32.
// Pipeline: register-staged double buffering -- global loads for K-step t+1 are issued before
// the MFMAs of step t and written to the other LDS buffer after them; one barrier per K-step.
// LDS rows are padded to 80 B (40 halves) so the 16 rows a lane group reads spread over banks.
// Workgroup ids are remapped so that the N-tiles sharing one activation row-block run on the
// same XCD (its 4 MiB L2 then serves the re-reads of that block).
#include "conv_igemm.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace rmr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


// v * rcp(1 + e^-v): the hardware reciprocal (1 ulp) instead of an IEEE division -- the epilogue's VALU
// work is not small beside a short K loop (48 values per lane per tile)
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

template <int WM, int WN, int MREP, int NREP, int BK, bool UT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int BM = WM * MREP * 16;
    constexpr int BN = WN * NREP * 16;
    // LDS row stride BK+16 halves: with 6 (BK=32) / 10 (BK=64) 16-byte slots per row the four
    // 16-lane groups of a ds_read_b128 (rows r, k-group g -> slot (r*S+g) mod 16) hit 16
    // distinct slots: conflict-free fragment reads (BK+8 was 2-way: SQ_LDS_BANK_CONFLICT = 50 %).
    constexpr int LDK = BK + 16;
    constexpr int CPR = BK / 8;     // 16-byte chunks per row of the K slice
    constexpr int RPI = 256 / CPR;  // rows staged per pass of the 256 threads
    constexpr int A_IT = BM / RPI;
    constexpr int B_IT = (BN + RPI - 1) / RPI;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BM % RPI == 0, "BM must be a multiple of the staging pass");
    static_assert(BK == 32 || BK == 64, "BK is 32 or 64");

    __shared__ __attribute__((aligned(16))) _Float16 As[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[2][BN * LDK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: logical id runs through the N-tiles of one M-block first, and
    // consecutive logical ids are dispatched to the same XCD (block b -> XCD b % 8).
    const int nt_count = a.Cout_pad / BN;
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = blockIdx.x & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int m0 = (lid / nt_count) * BM;
    const int n0 = (lid % nt_count) * BN;

    // Both operands are fetched with buffer loads: 32-bit byte offsets, and an offset past
    // num_records returns zeros -- that is how padding taps, the K tail and rows past M are
    // zero-filled without a branch or a select on the data.
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const auto wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;

    // ---- per-thread im2col bookkeeping (rows are fixed across the K loop) ----
    const int kc = tid % CPR;  // 16-byte chunk within the BK-wide K slice
    const int row0 = tid / CPR;
    int a_off[A_IT];        // byte offset of element (n, ih0, iw0, in_co + kc*8); may be "negative"
    unsigned a_mask[A_IT];  // bit (kh*KW + kw): that filter tap reads a real input pixel
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + row0 + i * RPI;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int ow = mm % a.Wo;
        const int t = mm / a.Wo;
        const int oh = t % a.Ho;
        const int n = t / a.Ho;
        const int ih0 = oh * a.stride - a.pad;
        const int iw0 = ow * a.stride - a.pad;
        a_off[i] = (((n * a.H + ih0) * a.W + iw0) * a.in_cs + a.in_co + (UT ? kc * 8 : 0)) * 2;
        unsigned mask = 0;
        if (ok) {
            for (int r = 0; r < a.KH; ++r)
                for (int c = 0; c < a.KW; ++c)
                    if ((unsigned)(ih0 + r) < (unsigned)a.H && (unsigned)(iw0 + c) < (unsigned)a.W)
                        mask |= 1u << (r * a.KW + c);
        }
        a_mask[i] = mask;
    }
    // Position of the K slice inside the filter window.  UT (Cin % BK == 0): a slice never
    // straddles two taps, so (kh, kw, ci) is wave-uniform and lives in scalar registers;
    // otherwise every thread tracks the tap of its own 16-byte chunk.
    int k_ci, k_kw, k_kh;
    {
        const int k = UT ? 0 : kc * 8;
        k_ci = k % a.Cin;
        const int t = k / a.Cin;
        k_kw = t % a.KW;
        k_kh = t / a.KW;
    }
    int w_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int r = row0 + i * RPI;
        w_off[i] = ((n0 + (r < BN ? r : 0)) * a.Kp + kc * 8) * 2;
    }

    u32x4 a_reg[A_IT], b_reg[B_IT];
#define RMR_LOAD_TILES(kt)                                                                         \
    {                                                                                              \
        const int delta = ((k_kh * a.W + k_kw) * a.in_cs + k_ci) * 2;                              \
        const unsigned tap_bit = k_kh < a.KH ? 1u << (k_kh * a.KW + k_kw) : 0u;                     \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                         \
            const unsigned off = (a_mask[i] & tap_bit) ? (unsigned)(a_off[i] + delta) : OOB;       \
            a_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);                  \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i)                                           \
            b_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(wt_rsrc, w_off[i], (kt) * BK * 2, 0); \
        k_ci += BK;                                                                                \
        while (k_ci >= a.Cin) {                                                                    \
            k_ci -= a.Cin;                                                                         \
            if (++k_kw == a.KW) {                                                                  \
                k_kw = 0;                                                                          \
                ++k_kh;                                                                            \
            }                                                                                      \
        }                                                                                          \
    }
#define RMR_STORE_TILES(buf)                                                                       \
    {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i)                                           \
            *(u32x4*)&As[buf][(row0 + i * RPI) * LDK + kc * 8] = a_reg[i];                         \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                         \
            const int r = row0 + i * RPI;                                                          \
            if (r < BN) *(u32x4*)&Bs[buf][r * LDK + kc * 8] = b_reg[i];                            \
        }                                                                                          \
    }

    floatx4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.Kp / BK;
    RMR_LOAD_TILES(0);
    RMR_STORE_TILES(0);
    __syncthreads();

    const int frag_row = lane & 15;
    const int frag_k = (lane >> 4) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) RMR_LOAD_TILES(kt + 1);
        __builtin_amdgcn_sched_barrier(0);  // loads are issued before, and wait after, the MFMAs

#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            half8 xf[MREP], wf[NREP];
#pragma unroll
            for (int i = 0; i < MREP; ++i)
                xf[i] = *(const half8*)&As[buf][((wm * MREP + i) * 16 + frag_row) * LDK + ks * 32 + frag_k];
#pragma unroll
            for (int j = 0; j < NREP; ++j)
                wf[j] = *(const half8*)&Bs[buf][((wn * NREP + j) * 16 + frag_row) * LDK + ks * 32 + frag_k];
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }

        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) RMR_STORE_TILES(buf ^ 1);
        __syncthreads();
    }
#undef RMR_LOAD_TILES
#undef RMR_STORE_TILES

    // ---- epilogue: bias, SiLU, residual, store 4 consecutive channels per lane ----
    const int px = lane & 15;
    const int cq = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = m0 + (wm * MREP + i) * 16 + px;
        if (m >= a.M) continue;
        long mpre = 0;  // the half-resolution pixel under output pixel m
        if (a.pre) {
            const int hw = a.Ho * a.Wo, img = m / hw, rem = m - img * hw, y = rem / a.Wo, x = rem - y * a.Wo;
            mpre = ((long)img * (a.Ho >> 1) + (y >> 1)) * (a.Wo >> 1) + (x >> 1);
        }
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            const int n = n0 + (wn * NREP + j) * 16 + cq;
            const float4 b = *(const float4*)(a.bias + n);
            float v[4] = {acc[i][j][0] + b.x, acc[i][j][1] + b.y, acc[i][j][2] + b.z,
                          acc[i][j][3] + b.w};
            if (a.pre) {
                const float4 t = *(const float4*)(a.pre + mpre * a.pre_cs + n);
                v[0] += t.x, v[1] += t.y, v[2] += t.z, v[3] += t.w;
            }
            if (a.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
            }
            if (a.res) {
                union {
                    uint2 u;
                    _Float16 h[4];
                } rr;
                rr.u = *(const uint2*)((const _Float16*)a.res + (long)m * a.res_cs + a.res_co + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rr.h[r];
            }
            if (a.out32) {
                *(float4*)(a.out32 + (long)m * a.out_cs + a.out_co + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                union {
                    uint2 u;
                    _Float16 h[4];
                } o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o.h[r] = (_Float16)v[r];
                *(uint2*)((_Float16*)a.out + (long)m * a.out_cs + a.out_co + n) = o.u;
            }
        }
    }
}

// ---- tile table ------------------------------------------------------------------------------

struct TileDef {
    int bm, bn, bk;
    void (*kernel)(const ConvArgs);     // any Cin
    void (*kernel_ut)(const ConvArgs);  // Cin % BK == 0: wave-uniform filter tap per K slice
};

#define TILE(WM, WN, MR, NR) \
    { WM * MR * 16, WN * NR * 16, 32, conv_igemm_kernel<WM, WN, MR, NR, 32, false>, conv_igemm_kernel<WM, WN, MR, NR, 32, true> }
#define TILE64(WM, WN, MR, NR) \
    { WM * MR * 16, WN * NR * 16, 64, conv_igemm_kernel<WM, WN, MR, NR, 64, false>, conv_igemm_kernel<WM, WN, MR, NR, 64, true> }

static const TileDef kTiles[] = {
    TILE(4, 1, 4, 6),  // 0: 256 x 96
    TILE(2, 2, 4, 3),  // 1: 128 x 96
    TILE(2, 2, 2, 3),  // 2:  64 x 96
    TILE(4, 1, 4, 3),  // 3: 256 x 48
    TILE(4, 1, 2, 3),  // 4: 128 x 48
    TILE(4, 1, 1, 3),  // 5:  64 x 48
    TILE(4, 1, 4, 4),  // 6: 256 x 64
    TILE(2, 2, 4, 2),  // 7: 128 x 64
    TILE(2, 2, 2, 2),  // 8:  64 x 64
    TILE(2, 2, 4, 4),  // 9: 128 x 128
    TILE(2, 2, 2, 4),  // 10: 64 x 128
    TILE(4, 1, 4, 2),  // 11: 256 x 32
    TILE(4, 1, 1, 2),  // 12:  64 x 32
    TILE(4, 1, 4, 1),  // 13: 256 x 16
    TILE(4, 1, 1, 1),  // 14:  64 x 16
    // BK = 64: half the barriers and staging bookkeeping per MFMA
    TILE64(2, 2, 4, 3),  // 15: 128 x 96
    TILE64(2, 2, 4, 4),  // 16: 128 x 128
    TILE64(2, 2, 4, 2),  // 17: 128 x 64
    TILE64(4, 1, 2, 3),  // 18: 128 x 48
    TILE64(4, 1, 4, 3),  // 19: 256 x 48
    TILE64(2, 2, 2, 3),  // 20:  64 x 96
    TILE64(2, 2, 2, 4),  // 21:  64 x 128
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

int conv_num_tiles() { return kNumTiles; }
ConvTile conv_tile(int id) { return ConvTile{kTiles[id].bm, kTiles[id].bn, kTiles[id].bk}; }

int conv_pick_tile(int M, int cout_pad, int num_cus) {
    // RMR_BK=32|64 restricts the choice (tuning experiments); default: BK = 64 where a tile exists
    static const int want_bk = [] {
        const char* e = std::getenv("RMR_BK");
        return e ? std::atoi(e) : 0;
    }();
    static const int want_bm = [] {
        const char* e = std::getenv("RMR_BM");
        return e ? std::atoi(e) : 0;
    }();
    // widest BN that divides Cout_pad (fewer re-reads of the activation tile) ...
    int bn = 16;
    for (int cand : {128, 96, 64, 48, 32, 16})
        if (cout_pad % cand == 0) {
            bn = cand;
            break;
        }
    // ... then the tallest BM that still yields >= 2 workgroups per CU; else the shortest.
    // Between equal shapes the deeper K slice wins.
    int best = -1, smallest = -1;
    auto better = [&](int t, int cur, bool want_tall) {
        if (cur < 0) return true;
        if (kTiles[t].bm != kTiles[cur].bm) return want_tall ? kTiles[t].bm > kTiles[cur].bm : kTiles[t].bm < kTiles[cur].bm;
        return kTiles[t].bk > kTiles[cur].bk;
    };
    for (int t = 0; t < kNumTiles; ++t) {
        if (kTiles[t].bn != bn) continue;
        if (want_bk && kTiles[t].bk != want_bk) continue;
        if (want_bm && kTiles[t].bm != want_bm) continue;
        const long blocks = (long)((M + kTiles[t].bm - 1) / kTiles[t].bm) * (cout_pad / bn);
        if (blocks >= 2L * num_cus && better(t, best, true)) best = t;
        if (better(t, smallest, false)) smallest = t;
    }
    if (best < 0 && smallest < 0 && want_bk) {  // no tile of that BK for this BN: lift the restriction
        for (int t = 0; t < kNumTiles; ++t)
            if (kTiles[t].bn == bn && better(t, smallest, false)) smallest = t;
    }
    return best >= 0 ? best : smallest;
}

void launch_conv(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile) {
    if (tile < 0 || tile >= kNumTiles) fail(RMR_ERR_INVALID_ARGUMENT, "conv: tile %d out of range", tile);
    const TileDef& t = kTiles[tile];
    if (a.Cout_pad % t.bn) fail(RMR_ERR_LOGIC, "conv: Cout_pad %d not a multiple of tile BN %d", a.Cout_pad, t.bn);
    if (a.Cin % 8 || a.in_cs % 8 || a.in_co % 8 || a.out_cs % 4 || a.out_co % 4 || a.Kp % t.bk || a.KH * a.KW > 32)
        fail(RMR_ERR_LOGIC, "conv: misaligned view (Cin %d in_cs %d in_co %d out_cs %d out_co %d Kp %d)",
             a.Cin, a.in_cs, a.in_co, a.out_cs, a.out_co, a.Kp);
    if (a.in_bytes == 0 || a.in_bytes > 0xf0000000ull || a.wt_bytes == 0)
        fail(RMR_ERR_LOGIC, "conv: buffer sizes not set or input view larger than 3.75 GiB");
    const int grid = ((a.M + t.bm - 1) / t.bm) * (a.Cout_pad / t.bn);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad +
                                (double)a.Cout_pad * a.K);
    // RMR_PROFILE_LAYERS=1: one profile entry per GEMM shape instead of one for the kernel
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d t%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, tile);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    (a.Cin % t.bk == 0 ? t.kernel_ut : t.kernel)<<<grid, 256, 0, stream>>>(a);
    RMR_HIP(hipGetLastError());
}

void launch_conv_auto(DeviceCtx& ctx, hipStream_t stream, const ConvArgs& a) {
    static const int mode = [] {
        const char* e = std::getenv("RMR_CONV");
        if (!e) return 0;
        return std::string(e) == "igemm" ? 1 : 2;
    }();
    const bool dma = mode != 1 && conv_dma_supported(a);
    if (dma)
        launch_conv_dma(ctx, stream, a, conv_dma_pick_tile(a.M, a.Cout_pad, ctx.num_cus));
    else
        launch_conv(ctx, stream, a, conv_pick_tile(a.M, a.Cout_pad, ctx.num_cus));
}

void pack_conv_weights(const float* w, int cout, int cin, int kh, int kw, int cin_pad, int cout_pad,
                       std::vector<__half>& out, int& K, int& Kp) {
    K = kh * kw * cin_pad;
    Kp = (K + 63) / 64 * 64;  // every tile's BK (32 or 64) divides it
    out.assign((size_t)cout_pad * Kp, __float2half(0.f));
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
            for (int r = 0; r < kh; ++r)
                for (int s = 0; s < kw; ++s)
                    out[(size_t)o * Kp + (size_t)(r * kw + s) * cin_pad + c] =
                        __float2half(w[(((size_t)o * cin + c) * kh + r) * kw + s]);
}

}  // namespace rmr
