// conv_g32.hip -- implicit-GEMM convolution for the layers conv_t32 cannot take (1x1, and 3x3 / stride 2),
// on the same skeleton: 32x32x16 MFMAs, an 8-wave workgroup per CU walking tiles, weights streamed as
// pre-packed LDS images, fragment reads half a stage ahead of the MFMAs and across the barrier.
//
// What differs is the pixel operand.  A stride-1 3x3 conv stages an input RANGE once per 32-channel chunk and
// takes its nine taps as row shifts; with stride 2 (or no taps at all) there is no such reuse inside a tile
// of a few hundred output pixels, so here one STAGE = one (tap, 32-channel chunk) and its ring slice holds both
// operands: the BM x 64-byte rows of the tile's pixels at that tap, gathered by the DMA itself (one 64-byte
// row per output pixel, padding and rows past M as out-of-range offsets, which the DMA turns into zeros),
// and the BN x 64-byte weight slice behind them.  That is 28 KiB per 24 MFMAs of a 256 x 192 tile, 36 bytes
// per clock and CU at full MFMA rate -- under what the L2 -> LDS path delivers (tools/microbench/lds_dma_rate.hip:
// 42 B/clk for 64-byte rows, 61 B/clk for contiguous KiB), which round 1's im2col kernel (conv_dma) never
// approached because its K loop was bound by issue slots, not bytes.
//
// LDS rows, fragment geometry, swizzle and weight image are conv_t32's (conv_t32.hip, head comment).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include "conv_t32_common.h"

namespace rmr {

namespace {

using namespace t32;

// WM x WN waves, MREP x NREP fragments of 32 x 32 per wave, T = KH * KW taps, R stages in the ring.
// PERSISTENT like conv_t32: the DMA stream runs R - 1 stages ahead of the MFMAs and does not stop at a tile's
// end, so the epilogue of a tile runs with the next tile's first stages in flight.
template <int WM, int WN, int MREP, int NREP, int T, int R>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv_g32_kernel(const ConvArgs a, const int n_tiles) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * MREP * 32;
    constexpr int BN = WN * NREP * 32;
    constexpr int NA = BM / 16, NB = BN / 16;   // DMA instructions per stage: pixel rows, weights
    constexpr int D = (NA + NB + NW - 1) / NW;  // per wave
    constexpr int DA = NA / NW;                 // of which pixel rows (slots 0 .. DA-1 of every wave)
    constexpr int SLICE = (BM + BN) * 64;       // one stage: [BM pixel rows][BN weight rows]
    constexpr int KW_ = T == 9 ? 3 : 1;
    constexpr unsigned OOB = 0xffff0000u;
    static_assert(T == 1 || T == 9, "1x1 or 3x3");
    static_assert(NA % NW == 0, "a DMA slot is either pixels or weights for all waves");
    static_assert(R >= 4 && R <= 6 && (R - 3) * D <= 63, "ring depth / vmcnt is 6 bits");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const unsigned lds0 = sgpr((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    constexpr int scratch_off = R * SLICE;   // where the DMA slots with nothing to fetch land (only if (NA + NB) % NW)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // tile vb -> (m0, n0); XCD-aware: the tiles of one XCD (vb & 7) are a contiguous range, n-tiles innermost
    const int nt_count = a.Cout_pad / BN;
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
    const int G = gridDim.x;  // a multiple of 8: vb & 7 is this workgroup's XCD for every tile it walks
    const auto tile_m0n0 = [&](int vb, int& m0, int& n0) {
        const int xcd = vb & 7;
        const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
        m0 = (lid / nt_count) * BM;
        n0 = (lid % nt_count) * BN;
    };
    int vb = blockIdx.x;
    if (vb >= n_tiles) return;
    int m0, n0;
    tile_m0n0(vb, m0, n0);

    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu),
                           sgpr(a.in_bytes), sgpr(0x00020000u)};
    const u32x4 wt_rsrc = {sgpr((unsigned)(size_t)a.wt_t32), sgpr((unsigned)((size_t)a.wt_t32 >> 32) & 0xffffu),
                           sgpr(a.wt_t32_bytes), sgpr(0x00020000u)};

    // ---- DMA constants of this lane ----------------------------------------------------------
    const int lrow = lane >> 2;                                   // row inside a 16-row DMA block
    const int lch = (lane & 3) ^ ((lrow >> 2) & 3);               // logical 16-byte chunk it fetches
    const int cs2 = a.in_cs * 2;
    const int in_cb = (a.in_co + lch * 8) * 2;
    const unsigned lane16 = (unsigned)lane * 16u;
    const int chunks = a.Cin / 32;
    const int total = chunks * T;
    const unsigned wstep = (unsigned)(a.Cout_pad / 16) * 1024u;   // bytes of one stage's slice of all channels
    const int W = a.W, H = a.H, S = a.stride, P = a.pad;
    const unsigned row_b = (unsigned)(W * cs2);                   // bytes of one input image row

    // pixel-row slot j of this wave is tile row (wave + NW * j) * 16 + lrow: byte offset of its pixel at tap (0, 0),
    // chunk 0 (meaningless where the tap mask is clear), and the taps that exist for it
    const auto decompose = [&](int m0_t, int j, int& base, unsigned& mask) {
        const int m = m0_t + (wave + NW * j) * 16 + lrow;
        const int ox = m % a.Wo, oy = (m / a.Wo) % a.Ho, n = m / (a.Wo * a.Ho);
        const int iy = oy * S - P, ix = ox * S - P;
        base = ((n * H + iy) * W + ix) * cs2 + in_cb;
        mask = 0;
        if (m < a.M) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int y = iy + t / KW_, x = ix + t % KW_;
                if (y >= 0 && y < H && x >= 0 && x < W) mask |= 1u << t;
            }
        }
    };
    // the tile the DMA stream is fetching (f*) and the one behind it (n*)
    int fbase[DA], nbase[DA];
    unsigned fmask[DA], nmask[DA];
#pragma unroll
    for (int j = 0; j < DA; ++j) decompose(m0, j, fbase[j], fmask[j]);

    // weight slot j (DA <= j < D): block q = wave + NW * j - NA of the slice, if the slice has that many
    bool s_wlive[D];
    unsigned s_dst[D], s_wsrc[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const int q = wave + NW * j;
        s_wlive[j] = q < NA + NB;
        s_dst[j] = q * 1024;
        s_wsrc[j] = (unsigned)(q - NA) * 1024u;
    }

    // the stream: stage fs of the tile whose channel-tile offset is w_tile is the next one to fetch
    unsigned fs = 0, fwoff = 0, fcoff = 0;   // stage, its weight offset (fs * wstep), its chunk's byte offset in a pixel
    unsigned w_tile = (unsigned)(n0 / 16) * 1024u;
    unsigned w_live = 1u;

    // the DMA instructions of the stream's current stage (tap tf of it known at compile time) into ring slot `dst`
    const auto issue_slot = [&](auto Jc, auto Tf, int dst) {
        constexpr int j = decltype(Jc)::value;
        constexpr int tf = decltype(Tf)::value;
        const unsigned lds = sgpr(lds0 + dst * SLICE);
        if constexpr (j < DA) {
            const unsigned tap_b = (unsigned)(tf / KW_) * row_b + (unsigned)((tf % KW_) * cs2);
            unsigned av = (unsigned)fbase[j] + tap_b + fcoff;
            const bool ok = (fmask[j] >> tf) & 1u;
            av = (ok && w_live) ? av : OOB;
            dma16s(in_rsrc, lds + s_dst[j], av, 0u);
        } else {
            const bool ok = s_wlive[j] && w_live;
            dma16s(wt_rsrc, ok ? lds + s_dst[j] : sgpr(lds0 + scratch_off), ok ? lane16 : OOB, sgpr(s_wsrc[j] + w_tile + fwoff));
        }
    };
    // ... and the step to the next stage; behind a tile's last stage comes the first one of the next tile.  Selects, not
    // branches: a stage must stay one basic block, or the scheduling pins do not hold the MFMAs in place.  `chunk_edge`:
    // the stage stepped to is tap 0 of a chunk (the only place where the chunk offset moves and the tile can change).
    unsigned w_tile_next = 0, next_live = 0;
    const auto advance = [&](auto Edge) {
        constexpr bool chunk_edge = decltype(Edge)::value;
        const unsigned wrap = 0u - (unsigned)(fs + 1 == (unsigned)total);   // all ones behind the tile's last stage
        fs = (fs + 1) & ~wrap;
        fwoff = (fwoff + wstep) & ~wrap;
        if constexpr (chunk_edge) {
            fcoff = (fcoff + 64u) & ~wrap;
            w_tile ^= (w_tile ^ w_tile_next) & wrap;
            w_live ^= (w_live ^ next_live) & wrap;
#pragma unroll
            for (int j = 0; j < DA; ++j) {
                fbase[j] ^= (fbase[j] ^ nbase[j]) & (int)wrap;
                fmask[j] ^= (fmask[j] ^ nmask[j]) & wrap;
            }
        }
    };

    // ---- fragment constants ------------------------------------------------------------------------
    const int fr = lane & 31, kq = lane >> 5;
    const int key16 = (kq ^ ((fr >> 2) & 3)) << 4;
    const int alane = (wm * MREP * 32 + fr) * 64 + key16;
    const int wlane = BM * 64 + (wn * NREP * 32 + fr) * 64 + key16;
    const auto lds16 = [&](int off) { return *(const half8*)(smem + off); };

    // ---- cold start (the first tile of this workgroup only): stages 0 .. R-2; a tile has at least R stages, so the
    // stream cannot wrap here and the next tile's state is not looked at yet
#pragma unroll
    for (int j = 0; j < DA; ++j) nbase[j] = 0, nmask[j] = 0;
    static_for<0, R - 1>([&](auto Sc) {
        constexpr int s = decltype(Sc)::value;
        static_for<0, D>([&](auto Jc) { issue_slot(Jc, tap_c<s % T>{}, s); });
        advance(std::integral_constant<bool, (s + 1) % T == 0>{});
    });
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    constexpr int NM = MREP * NREP;   // MFMAs per K-step
    int slot = 0;                     // ring slot of the stage being computed

    for (;;) {
        // ---- this tile and the next one -------------------------------------------------------------
        const int vbn = vb + G;
        const bool has_next = vbn < n_tiles;
        int m0n = 0, n0n = 0;
        if (has_next) tile_m0n0(vbn, m0n, n0n);
        w_tile_next = (unsigned)(n0n / 16) * 1024u;
        next_live = (unsigned)has_next;
#pragma unroll
        for (int j = 0; j < DA; ++j) decompose(m0n, j, nbase[j], nmask[j]);

        floatx16 acc[MREP][NREP];
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // fragments of (stage 0, K-step 0): waited for before the last barrier
        half8 xa[MREP], wa[NREP], xb[MREP], wb[NREP];
        int cur = slot * SLICE;
#pragma unroll
        for (int i = 0; i < MREP; ++i) xa[i] = lds16(cur + alane + i * 2048);
#pragma unroll
        for (int j = 0; j < NREP; ++j) wa[j] = lds16(cur + wlane + j * 2048);

        for (int cc = 0; cc < chunks; ++cc) {
            // One stage = 2 NM MFMAs (K-step 0, then K-step 1) with everything else placed by hand into the gaps behind
            // them: the K-step 1 fragments, the D DMA instructions of the stage R - 1 ahead (into the slot the last
            // stage left), the next stage's K-step 0 fragments (legal before the barrier that opens it: that slice was
            // waited for one stage ago).
            const auto stage = [&](auto Tc) {
                constexpr int t = decltype(Tc)::value;
                constexpr int tf = (t + R - 1) % T;   // tap of the stage the stream is at
                const int slot_w = slot == 0 ? R - 1 : slot - 1;
                int nxt = 0;
                __builtin_amdgcn_s_barrier();
                const auto filler0 = [&](auto Fc) {
                    constexpr int f = decltype(Fc)::value;
                    if constexpr (f == 0) {
#pragma unroll
                        for (int i = 0; i < MREP; ++i) xb[i] = lds16((cur + alane + i * 2048) ^ 32);
                    } else if constexpr (f == 1) {
#pragma unroll
                        for (int j = 0; j < NREP; ++j) wb[j] = lds16((cur + wlane + j * 2048) ^ 32);
                    } else if constexpr (f < 2 + D) {
                        issue_slot(std::integral_constant<int, f - 2>{}, tap_c<tf>{}, slot_w);
                    } else {
                        const int slot_n = slot + 1 == R ? 0 : slot + 1;
                        nxt = slot_n * SLICE;
                        slot = slot_n;
                    }
                };
                static_for<0, NM>([&](auto Kc) {
                    constexpr int k = decltype(Kc)::value;
                    acc[k / NREP][k % NREP] =
                        __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[k % NREP], xa[k / NREP], acc[k / NREP][k % NREP], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, D + 3>([&](auto Fc) {
                        if constexpr (decltype(Fc)::value * NM / (D + 3) == k) filler0(Fc);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                static_for<0, NM>([&](auto Kc) {
                    constexpr int k = decltype(Kc)::value;
                    acc[k / NREP][k % NREP] =
                        __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[k % NREP], xb[k / NREP], acc[k / NREP][k % NREP], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (k == 0) {
#pragma unroll
                        for (int i = 0; i < MREP; ++i) xa[i] = lds16(nxt + alane + i * 2048);
                    }
                    if constexpr (k == (NM > 1 ? 1 : 0)) {
#pragma unroll
                        for (int j = 0; j < NREP; ++j) wa[j] = lds16(nxt + wlane + j * 2048);
                        cur = nxt;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // DMAs issued R - 3 stages ago (and earlier) have landed -- and, behind an epilogue, its stores, which are
                // older than this stage's DMAs (loads and stores retire in order)
                wait_vm<(R - 3) * D>();
                advance(std::integral_constant<bool, (tf + 1) % T == 0>{});
            };
            if constexpr (T == 9) {
                stage(tap_c<0>{});
                stage(tap_c<1>{});
                stage(tap_c<2>{});
                stage(tap_c<3>{});
                stage(tap_c<4>{});
                stage(tap_c<5>{});
                stage(tap_c<6>{});
                stage(tap_c<7>{});
                stage(tap_c<8>{});
            } else {
                stage(tap_c<0>{});
            }
        }

        // ---- epilogue; the next tile's first stages are in flight meanwhile
        epilogue<MREP, NREP, 0, false, false>(a, acc, smem, 0, 0, m0, n0, wm, wn, lane);

        if (!has_next) break;
        vb = vbn;
        m0 = m0n;
        n0 = n0n;
    }
    wait_vm<0>();
}

struct G32Tile {
    int bm, bn, threads, taps, ring, wgs_per_cu;
    void (*kernel)(const ConvArgs, int);
};

#define G32(WM, WN, MR, NR, T, R, WPC) \
    { WM * MR * 32, WN * NR * 32, WM * WN * 64, T, R, WPC, conv_g32_kernel<WM, WN, MR, NR, T, R> }

const G32Tile kG32Tiles[] = {
    G32(4, 2, 2, 3, 1, 5, 1),   // 0: 1x1, 256 x 192
    G32(8, 1, 2, 3, 1, 4, 1),   // 1: 1x1, 512 x 96
    G32(4, 2, 2, 3, 9, 5, 1),   // 2: 3x3, 256 x 192
    G32(8, 1, 2, 3, 9, 4, 1),   // 3: 3x3, 512 x 96
    G32(4, 2, 2, 2, 1, 6, 1),   // 4: 1x1, 256 x 128
    G32(4, 2, 2, 2, 9, 6, 1),   // 5: 3x3, 256 x 128
    // two four-wave workgroups per CU (80 KiB each: a ring of four 20 KiB stages, every DMA slot live): one's
    // epilogue under the other's K loop
    G32(2, 2, 2, 3, 1, 4, 2),   // 6: 1x1, 128 x 192
    G32(2, 2, 2, 3, 9, 4, 2),   // 7: 3x3, 128 x 192
};
constexpr int kNumG32Tiles = sizeof(kG32Tiles) / sizeof(kG32Tiles[0]);

// the scratch KiB behind the ring exists only where a wave has a DMA slot without a block
int g32_lds_bytes(const G32Tile& t) {
    const bool dead_slots = ((t.bm + t.bn) / 16) % (t.threads / 64) != 0;
    return t.ring * (t.bm + t.bn) * 64 + (dead_slots ? 1024 : 0);
}

}  // namespace

int conv_g32_num_tiles() { return kNumG32Tiles; }
ConvTile conv_g32_tile(int id) { return ConvTile{kG32Tiles[id].bm, kG32Tiles[id].bn, 32}; }

bool conv_g32_supported(const ConvArgs& a, int tile) {
    if (a.KH != a.KW || (a.KH != 1 && a.KH != 3) || a.Cin % 32 || a.Cin < 32) return false;
    if (a.pre || a.in_slab_c || a.out_slab_c || !a.wt_t32) return false;
    if (tile < 0) return true;
    const G32Tile& t = kG32Tiles[tile];
    // the stream's cold start and its wrap into the next tile assume a tile has at least a ring of stages
    return t.taps == a.KH * a.KW && a.Cout_pad % t.bn == 0 && a.Cin / 32 * t.taps >= t.ring;
}

void launch_conv_g32(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile) {
    if (tile < 0 || tile >= kNumG32Tiles) fail(RMR_ERR_INVALID_ARGUMENT, "conv_g32: tile %d out of range", tile);
    if (!conv_g32_supported(a, tile)) fail(RMR_ERR_LOGIC, "conv_g32: layer not supported by tile %d", tile);
    const G32Tile& t = kG32Tiles[tile];
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 4 || a.out_co % 4) fail(RMR_ERR_LOGIC, "conv_g32: misaligned view");
    if (a.in_bytes == 0 || a.in_bytes > 0xf0000000ull || a.wt_t32_bytes == 0)
        fail(RMR_ERR_LOGIC, "conv_g32: buffer sizes not set or input view larger than 3.75 GiB");
    static std::once_flag once;
    std::call_once(once, [] {
        for (const G32Tile& d : kG32Tiles)
            (void)hipFuncSetAttribute((const void*)d.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int lds = g32_lds_bytes(t);
    const int n_tiles = ((a.M + t.bm - 1) / t.bm) * (a.Cout_pad / t.bn);
    // persistent: wgs_per_cu workgroups per CU (a multiple of 8: a workgroup stays on its XCD), each walks tiles
    const int grid = std::min((n_tiles + 7) / 8 * 8, ctx.num_cus * t.wgs_per_cu);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d m%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, tile);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    t.kernel<<<grid, t.threads, lds, stream>>>(a, n_tiles);
    RMR_HIP(hipGetLastError());
}

}  // namespace rmr
