// conv_t32f8.hip -- the fp8 form of conv_t32 (BASELINE configs[4]: fp8-MFMA weights, batch 256).
//
// Same persistent tile walker, halo staging, LDS-image weights and lane masks as conv_t32.hip; what changes:
//   * operands are OCP e4m3: weights quantised on the host with one scale per output channel
//     (pack_conv_weights_t32f8), activations with unit scale (a SiLU output lies in [-0.28, a few tens];
//     e4m3 spans 2^-9 .. 448) by quant_f8_kernel below; the product is rescaled in the epilogue;
//   * an LDS row is still 64 bytes, now 64 channels; a lane of a fragment read takes 32 of them (two
//     ds_read_b128) and ONE v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales consumes them: twice the
//     FLOPs per pipe cycle of the f16 form and -- measured, tools/microbench/mfma_fp8_power.hip -- 3.9 PFLOP/s
//     at the power limit on random operands where f16 reaches 1.65;
//   * a tap is one K-step, so the read-ahead is organised per fragment: the weight fragments are single
//     buffered and re-read for the next tap as soon as their last MFMA of this tap has issued (MFMAs run
//     weight-fragment-major), the pixel fragments are double buffered.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>

#include "conv_igemm.h"
#include "conv_t32_common.h"

namespace rmr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int intx8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ float silu_t(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// LDS-DMA: 64 lanes x 16 bytes land at lds_addr + lane * 16; source = rsrc base + voff + soff
__device__ __forceinline__ void dma16s(u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

template <int T>
using tap_c = std::integral_constant<int, T>;

template <int K, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        static_for<K + 1, N>(f);
    }
}

// WM x WN waves, MREP x NREP fragments of 32 x 32 per wave, A_SLOTS input-range DMA instructions per
// tap (the first 11 - R taps of a 64-channel chunk carry the next chunk's range), R weight slices in the ring.
// EPI: 0 = results leave through v_permlane32_swap pairs as 16-byte stores (32 contiguous bytes per pixel),
//      1 = through a per-wave LDS stage as whole rows (NREP * 64 contiguous bytes per pixel).
//
// PERSISTENT: the grid is at most a few workgroups per CU and a workgroup walks tiles vb = b, b + G, ...
// The DMA stream does not stop at a tile's end: the last R - 1 taps of a tile fetch the first weight
// slices of the next one and its last chunk fetches the next tile's first input range, so only the very
// first tile of a workgroup pays a cold start, and the epilogue of a tile runs while the next tile's
// operands are already in flight.
template <int WM, int WN, int MREP, int NREP, int A_SLOTS, int R, int EPI, int WPC>
__global__ __launch_bounds__(WM* WN * 64, WPC * WM * WN / 4) void conv_t32f8_kernel(const ConvArgs a, const int a_rows, const int n_tiles, const int stagger) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * MREP * 32;
    constexpr int BN = WN * NREP * 32;
    constexpr int NB = BN / 16;                 // weight DMA instructions per tap
    constexpr int SLOTS = NB + A_SLOTS;         // DMA instructions per tap (workgroup)
    constexpr int D = (SLOTS + NW - 1) / NW;    // per wave
    constexpr int SLOT_BYTES = BN * 64;         // one (chunk, tap) weight slice
    constexpr int ATAPS = 11 - R;               // taps that carry input-range blocks
    constexpr int STG_PITCH = NREP * 64 + 16;   // bytes per pixel row of the epilogue stage
    constexpr unsigned OOB = 0xffff0000u;
    static_assert(R >= 4 && R <= 6, "ring depth");
    static_assert((R - 3) * D <= 63, "vmcnt is 6 bits");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const unsigned lds0 = sgpr((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const int a_buf_bytes = a_rows * 64;
    const int ring_base = 2 * a_buf_bytes;
    const int zero_off = ring_base + R * SLOT_BYTES;   // 64 zero bytes, head of the scratch KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int stg_base = zero_off + 1024 + wave * 32 * STG_PITCH;
    const int bias_off = zero_off + 1024 + (EPI ? NW * 32 * STG_PITCH : 0);   // Cout_pad bias floats, then Cout_pad scales

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
    const int W = a.W;
    const int npix = a.M;        // stride 1: input and output pixels share the linear index

    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in8), sgpr((unsigned)((size_t)a.in8 >> 32) & 0xffffu),
                           sgpr(a.in8_bytes), sgpr(0x00020000u)};
    const u32x4 wt_rsrc = {sgpr((unsigned)(size_t)a.wt8), sgpr((unsigned)((size_t)a.wt8 >> 32) & 0xffffu),
                           sgpr(a.wt8_bytes), sgpr(0x00020000u)};

    // ---- DMA constants of this lane ----------------------------------------------------------
    const int lrow = lane >> 2;                                   // row inside a 16-row DMA block
    const int lch = (lane & 3) ^ ((lrow >> 2) & 3);               // logical 16-byte chunk it fetches
    const unsigned cs2 = (unsigned)a.in8_cs;                      // bytes per pixel of the e4m3 activations
    const unsigned in_cb = (unsigned)(lch * 16);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int na = a_rows / 16;                                   // input-range DMA blocks per chunk
    const int chunks = (a.Cin + 63) / 64;   // a partial last chunk is zero padded in both operands
    const int total = chunks * 9;
    const unsigned wstep = (unsigned)(a.Cout_pad / 16) * 1024u;   // bytes of one (chunk, tap) slice of all channels
    const unsigned scratch = sgpr(lds0 + zero_off);

    if (tid < 4) *(u32x4*)(smem + zero_off + tid * 16) = u32x4{0, 0, 0, 0};
    for (int i = tid; i < a.Cout_pad; i += NW * 64) {
        *(float*)(smem + bias_off + i * 4) = a.bias[i];
        *(float*)(smem + bias_off + (a.Cout_pad + i) * 4) = a.wscale[i];
    }

    // byte offset of this lane's piece of input block ia of a tile whose LDS row 0 is pixel lo, channel chunk cc
    const auto in_off = [&](int lo_l, int ia, int cc) {
        const int p = min(max(lo_l + ia * 16, 0), npix - 1);  // out-of-range pixels are only ever read by masked taps
        return __umul24((unsigned)p, cs2) + in_cb + (unsigned)cc * 64u;
    };

    // DMA slot j of this wave is q = wave + NW * j: a weight block (q < NB) or an input-range block, for the
    // whole kernel -- the role is chosen once, the issue code in the K loop has no branches
    bool s_isw[D];
    u32x4 s_rsrc[D];
    unsigned s_wdst[D], s_wsrc[D];
    int s_aidx[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const int q = wave + NW * j;
        s_isw[j] = q < NB;
        s_rsrc[j] = NW * (j + 1) <= NB ? wt_rsrc : NW * j >= NB ? in_rsrc : (s_isw[j] ? wt_rsrc : in_rsrc);
        s_wdst[j] = lds0 + ring_base + q * 1024;
        s_wsrc[j] = (unsigned)q * 1024u;
        s_aidx[j] = q < SLOTS ? q - NB : 1 << 20;   // a surplus slot (D * NW > SLOTS) fetches nothing
    }

    // Two workgroups share a CU so that one's epilogue (two transcendentals per output value: a third of the
    // MFMA time of a K = 864 tile) runs under the other's K loop -- which only happens when they are out of
    // phase.  Launched together and walking equal tiles they would stay in lockstep for the whole launch, so the
    // workgroup in the CU's odd slot (HW_ID.TG_ID, the barrier resource it was given) starts half a tile late.
    if (stagger > 0) {
        const unsigned hw_id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 16 << 6 | 4);   // HW_REG_HW_ID[19:16] = TG_ID
        if (hw_id & 1)
            for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(64);   // 64 x 64 cycles each
    }

    // ---- cold start: the whole input range of chunk 0, weight slices 0 .. R-2 of the first tile -----------
    {
        const int pl0 = m0 - W - 1 + lrow;
        for (int ia = wave; ia < na; ia += NW) dma16s(in_rsrc, sgpr(lds0 + ia * 1024), in_off(pl0, ia, 0), 0u);
#pragma unroll
        for (int s = 0; s < R - 1; ++s)
            for (int q = wave; q < NB; q += NW)
                dma16s(wt_rsrc, sgpr(lds0 + ring_base + s * SLOT_BYTES + q * 1024), s < total ? lane16 : OOB,
                       sgpr((unsigned)s * wstep + (unsigned)(n0 / 16 + q) * 1024u));
    }

    // ---- fragment constants ------------------------------------------------------------------------
    const int fr = lane & 31, kq = lane >> 5;
    const int a_row0 = wm * MREP * 32 + fr + W + 1;   // LDS row of the centre tap of fragment 0
    int zsel[MREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i) zsel[i] = zero_off - i * 2048;
    const int wlane = ring_base + (wn * NREP * 32 + fr) * 64 + (((2 * kq) ^ ((fr >> 2) & 3)) << 4);
    // a lane's 32 channels of a row: the 16-byte chunks 2 kq and 2 kq + 1 (slots c ^ key: 16 bytes apart)
    const auto lds32 = [&](int off) {
        typedef int intx4 __attribute__((ext_vector_type(4)));
        const intx4 lo4 = *(const intx4*)(smem + off);
        const intx4 hi4 = *(const intx4*)(smem + (off ^ 16));
        return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // address of fragment 0's first chunk for tap t, in input buffer `abuf`
    const auto a_addr = [&](int abuf, int t) {
        const int row = a_row0 + (t / 3 - 1) * W + (t % 3 - 1);
        return abuf + row * 64 + (((2 * kq) ^ ((row >> 2) & 3)) << 4);
    };

    wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    constexpr int NM = MREP * NREP;   // MFMAs per K-step
    // the K loop's scalar state, as in conv_t32.hip (round 6): byte offsets instead of slot numbers, and the weight stream
    // crosses into the next tile behind the compile-time tap 9 - R of a tile's last chunk -- no slice counter, no per-tap selects
    int woff = 0;                               // ring slot of the tap being computed, as a byte offset
    int woff_prev = (R - 1) * SLOT_BYTES;       // the slot the previous tap left: where this tap's weight DMAs land
    int abuf = 0;                               // input buffer of the chunk being computed
    unsigned wsoff = (unsigned)(n0 / 16) * 1024u + (unsigned)(R - 1) * wstep;   // the next slice to fetch
    unsigned wv = lane16;                       // the weight DMAs' lane offset: OOB once the stream has run past the last tile

    for (;;) {
        // ---- this tile and the next one -------------------------------------------------------------
        const int vbn = vb + G;
        const bool has_next = vbn < n_tiles;
        int m0n = 0, n0n = 0;
        if (has_next) tile_m0n0(vbn, m0n, n0n);
        const int pl = m0 - W - 1 + lrow, pln = m0n - W - 1 + lrow;
        const unsigned w_tile_next = (unsigned)(n0n / 16) * 1024u;
        // valid taps of this lane's pixel, as four lane masks per fragment (rows past M compute garbage
        // that is never stored: an MFMA column is one pixel)
        bool up[MREP], dn[MREP], lf[MREP], rt[MREP];
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            const int m = m0 + (wm * MREP + i) * 32 + fr;
            const int x = m % W, y = (m / W) % a.H;
            up[i] = y > 0;
            dn[i] = y < a.H - 1;
            lf[i] = x > 0;
            rt[i] = x < W - 1;
        }
        floatx16 acc[MREP][NREP];
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // fragments of tap 0: the slice and the range were waited for before the last barrier
        intx8 X[MREP], Xn[MREP], Wf[NREP];   // pixel fragments of this tap and of the next one; weight fragments (rolling)
        int wcur = wlane + woff;
        {
            const int at = a_addr(abuf, 0);
#pragma unroll
            for (int i = 0; i < MREP; ++i) X[i] = lds32(((up[i] && lf[i]) ? at : zsel[i]) + i * 2048);
#pragma unroll
            for (int j = 0; j < NREP; ++j) Wf[j] = lds32(wcur + j * 2048);
        }
        int at_n = a_addr(abuf, 1);   // where the next tap's pixel fragments are (computed one tap ahead)

        for (int cc = 0; cc < chunks; ++cc) {
            const int abuf_next = a_buf_bytes - abuf;
            // the range fetched during this chunk: the next chunk of this tile, or chunk 0 of the next tile
            const bool in_tile = cc + 1 < chunks;
            const bool a_live = in_tile || has_next;
            const int a_pl = in_tile ? pl : pln;
            const int a_cc = in_tile ? cc + 1 : 0;
            const int na_live = a_live ? na : 0;                       // blocks of the range fetched during this chunk
            const unsigned a_base = lds0 + (unsigned)abuf_next;        // where they land
            // One tap = NM MFMAs of 64 pipe cycles, weight-fragment-major: fragment j serves MFMAs j MREP .. and is
            // re-read for the next tap behind its last one; the next tap's pixel fragments and the DMA issue ride
            // behind the others.  Everything read for tap t + 1 during tap t was waited for one tap ago.
            const auto tap = [&](auto T) {
                constexpr int t = decltype(T)::value;
                constexpr int tn = (t + 1) % 9;
                const int woff_n = woff + SLOT_BYTES == R * SLOT_BYTES ? 0 : woff + SLOT_BYTES;
                const int wnext = wlane + woff_n;
                __builtin_amdgcn_s_barrier();
                static_for<0, NM>([&](auto Kc) {
                    constexpr int k = decltype(Kc)::value;
                    constexpr int j = k / MREP, i = k % MREP;
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Wf[j], X[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                              0x7f7f7f7f);
                    __builtin_amdgcn_sched_barrier(0);
                    // the next tap's pixel fragment i behind MFMA i (tap 0 of the next chunk after tap 8; after the
                    // tile's last tap they are read in vain: the next tile's lane masks are not known here)
                    if constexpr (k < MREP) {
                        constexpr int dy = tn / 3 - 1, dx = tn % 3 - 1;
                        const bool v = (dy < 0 ? up[k] : dy > 0 ? dn[k] : true) && (dx < 0 ? lf[k] : dx > 0 ? rt[k] : true);
                        Xn[k] = lds32((v ? at_n : zsel[k]) + k * 2048);
                    }
                    // weight fragment j has served its last MFMA of this tap: the next tap's takes its registers
                    if constexpr (i == MREP - 1) Wf[j] = lds32(wnext + j * 2048);
                    // DMA slot d behind MFMA 1 + d (when there are that many), the addresses behind the last but one
                    static_for<0, D>([&](auto Dc) {
                        constexpr int d = decltype(Dc)::value;
                        if constexpr ((NM > D + 1 ? 1 + d : d * NM / D) == k) {
                            constexpr bool all_w = NW * (d + 1) <= NB, all_a = NW * d >= NB;
                            constexpr bool a_tap = t < ATAPS;
                            const unsigned w_lds = s_wdst[d] + woff_prev, w_soff = s_wsrc[d] + wsoff;
                            const int ia = t * A_SLOTS + s_aidx[d];
                            const bool alive = a_tap && ia < na_live;
                            const unsigned a_lds = alive ? a_base + ia * 1024 : scratch;
                            if constexpr (all_w) {
                                dma16s(wt_rsrc, sgpr(w_lds), wv, sgpr(w_soff));
                            } else if constexpr (all_a) {
                                if constexpr (a_tap) {
                                    unsigned av = in_off(a_pl, ia, a_cc);
                                    asm volatile("" : "+v"(av));   // computed unconditionally: a branch around it would split the tap's basic block
                                    dma16s(in_rsrc, sgpr(a_lds), alive ? av : OOB, 0u);
                                }
                            } else {
                                const bool isw = s_isw[d];
                                unsigned av = in_off(a_pl, ia, a_cc);
                                asm volatile("" : "+v"(av));   // computed unconditionally: a branch around it would split the tap's basic block
                                av = (a_tap && alive) ? av : OOB;
                                dma16s(s_rsrc[d], sgpr(isw ? w_lds : a_lds), isw ? wv : av, sgpr(isw ? w_soff : 0u));
                            }
                        }
                    });
                    if constexpr (k == NM - 1) {
                        constexpr int t2 = (t + 2) % 9;
                        at_n = a_addr(t + 2 >= 9 ? abuf_next : abuf, t2);
                        woff_prev = woff;
                        woff = woff_n;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                constexpr int pending = [] {
                    int n = 0;
                    for (int k = 0; k < R - 3; ++k) {
                        const int tt = (t - k + 9) % 9;
                        for (int j = 0; j < D; ++j) n += (NW * j >= NB && tt >= ATAPS) ? 0 : 1;
                    }
                    return n;
                }();
                wait_vm<pending>();
#pragma unroll
                for (int i = 0; i < MREP; ++i) X[i] = Xn[i];   // renamed away inside the unrolled chunk
                // (selects, not a branch: a tap must stay one basic block, or the scheduling pins above do not hold
                // the MFMAs in place and the compiler sinks them towards the end of the chunk)
                if constexpr (t == 9 - R) {
                    wsoff = in_tile ? wsoff + wstep : w_tile_next;
                    wv = in_tile ? wv : (has_next ? lane16 : OOB);
                } else {
                    wsoff += wstep;
                }
            };
            tap(tap_c<0>{});
            tap(tap_c<1>{});
            tap(tap_c<2>{});
            tap(tap_c<3>{});
            tap(tap_c<4>{});
            tap(tap_c<5>{});
            tap(tap_c<6>{});
            tap(tap_c<7>{});
            tap(tap_c<8>{});
            abuf = abuf_next;
        }

        // ---- epilogue: bias, SiLU, residual; a lane holds 4 x 4 consecutive channels of one pixel per fragment.
        // The next tile's first slices and input range are in flight meanwhile.
        const int cq = kq * 4;
        const bool wide = !a.out32 && ((a.out_cs | a.out_co) & 7) == 0;   // 16-byte stores need 8-channel alignment
        // the usual case (SiLU, no e4m3 copy of the output): the shared epilogue -- every load ahead of the first store,
        // bias and scales from LDS (conv_t32_common.h)
        const bool fast = wide && a.act && (!a.out8 || EPI == 0) && (!a.res || (NREP < 4 && ((a.res_cs | a.res_co) & 7) == 0));
        if (fast && a.out8) {
            // the output (also / only) as the next e4m3 layer's input, written here instead of by a quantiser pass
            if constexpr (EPI == 0) {
                if (a.out8_only)
                    t32::epilogue_wide<MREP, NREP, 0, false, true, false, true, 0, 1, 1>(a, acc, smem, stg_base, bias_off, m0, n0, wm, wn, lane);
                else if (NREP < 4 && a.res)
                    t32::epilogue_wide<MREP, NREP, 0, (NREP < 4), true, false, true, 0, 1, 2>(a, acc, smem, stg_base, bias_off, m0, n0, wm, wn, lane);
                else
                    t32::epilogue_wide<MREP, NREP, 0, false, true, false, true, 0, 1, 2>(a, acc, smem, stg_base, bias_off, m0, n0, wm, wn, lane);
            }
        } else if (fast) {
            if (NREP < 4 && a.res)
                t32::epilogue_wide<MREP, NREP, EPI, (NREP < 4), true, false, true>(a, acc, smem, stg_base, bias_off, m0, n0, wm, wn, lane);
            else
                t32::epilogue_wide<MREP, NREP, EPI, false, true, false, true>(a, acc, smem, stg_base, bias_off, m0, n0, wm, wn, lane);
        } else
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            const int m = m0 + (wm * MREP + i) * 32 + fr;
            if (!wide) {
                if (m >= a.M) continue;
#pragma unroll
                for (int j = 0; j < NREP; ++j)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int n = n0 + (wn * NREP + j) * 32 + gq * 8 + cq;
                        const float4 b = *(const float4*)(a.bias + n);
                        const float4 sc = *(const float4*)(a.wscale + n);
                        float v[4] = {acc[i][j][gq * 4 + 0] * sc.x + b.x, acc[i][j][gq * 4 + 1] * sc.y + b.y, acc[i][j][gq * 4 + 2] * sc.z + b.z,
                                      acc[i][j][gq * 4 + 3] * sc.w + b.w};
                        if (a.act) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = silu_t(v[r]);
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
                continue;
            }
            // f16 output in 16-byte pieces.  Channel groups gq and gq + 1 of a lane pair (l, l + 32) hold
            // channels 8 gq + {0..3 | 4..7} and 8 gq + 8 + {0..3 | 4..7}: one v_permlane32_swap per dword
            // gives the lower lane all eight channels of group gq and the upper lane those of group gq + 1.
#pragma unroll
            for (int j = 0; j < NREP; ++j) {
                int q8[2] = {0, 0};   // the e4m3 bytes of group pair 0, until group pair 1 completes the 32-channel block
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const int nb = n0 + (wn * NREP + j) * 32 + gp * 16;     // first channel of the pair of groups
                    float v[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 b = *(const float4*)(a.bias + nb + h * 8 + cq);
                        const float4 sc = *(const float4*)(a.wscale + nb + h * 8 + cq);
                        const int r0 = (gp * 2 + h) * 4;
                        v[h * 4 + 0] = acc[i][j][r0 + 0] * sc.x + b.x;
                        v[h * 4 + 1] = acc[i][j][r0 + 1] * sc.y + b.y;
                        v[h * 4 + 2] = acc[i][j][r0 + 2] * sc.z + b.z;
                        v[h * 4 + 3] = acc[i][j][r0 + 3] * sc.w + b.w;
                    }
                    if (a.act) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = silu_t(v[r]);
                    }
                    const int nl = nb + kq * 8;   // the eight channels this lane ends up with
                    union {
                        uint4 u;
                        _Float16 h[8];
                        unsigned w[4];
                    } o;
                    if (a.res) {
                        // the shortcut is added in f32 before the one rounding, so the values are exchanged as f32
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[r]), __float_as_uint(v[4 + r]), false, false);
                            v[r] = __uint_as_float(sw[0]);
                            v[4 + r] = __uint_as_float(sw[1]);
                        }
                        // lower lane: v[0..3] own group gq, v[4..7] the upper lane's group gq; upper lane: v[0..3] the lower
                        // lane's group gq + 1, v[4..7] own -- in both cases channels nl .. nl + 7 in order
                        union {
                            uint4 u;
                            _Float16 h[8];
                        } rr;
                        if (m < a.M) rr.u = *(const uint4*)((const _Float16*)a.res + (long)m * a.res_cs + a.res_co + nl);
#pragma unroll
                        for (int r = 0; r < 8; ++r) o.h[r] = (_Float16)(v[r] + (float)rr.h[r]);
                    } else {
                        union {
                            uint2 u;
                            _Float16 h[4];
                            unsigned w[2];
                        } lo2, hi2;
#pragma unroll
                        for (int r = 0; r < 4; ++r) lo2.h[r] = (_Float16)v[r], hi2.h[r] = (_Float16)v[4 + r];
                        const auto s0 = __builtin_amdgcn_permlane32_swap(lo2.w[0], hi2.w[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(lo2.w[1], hi2.w[1], false, false);
                        o.w[0] = s0[0];
                        o.w[1] = s1[0];
                        o.w[2] = s0[1];
                        o.w[3] = s1[1];
                    }
                    if (a.out8) {
                        // the same eight values once more as e4m3 (rounded from the f16 that is stored: what a
                        // quantiser pass over the stored tensor would write).  Lane pair (l, l + 32) holds channels
                        // [0..7 | 8..15] of the group pair gp = 0 and [16..23 | 24..31] of gp = 1: one more half
                        // exchange gives the lower lane bytes 0..15 and the upper lane bytes 16..31 of the 32-channel
                        // block -- one 16-byte store per lane and fragment instead of two of 8
                        const auto sat = [](_Float16 v) { return __builtin_amdgcn_fmed3f((float)v, -448.f, 448.f); };
                        int w0 = 0, w1 = 0;
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(o.h[0]), sat(o.h[1]), w0, false);
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(o.h[2]), sat(o.h[3]), w0, true);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(o.h[4]), sat(o.h[5]), w1, false);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(o.h[6]), sat(o.h[7]), w1, true);
                        if (gp == 0) {
                            q8[0] = w0, q8[1] = w1;
                        } else {
                            const auto s0 = __builtin_amdgcn_permlane32_swap((unsigned)q8[0], (unsigned)w0, false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap((unsigned)q8[1], (unsigned)w1, false, false);
                            // lower lane: own gp 0 (0..7), the upper lane's gp 0 (8..15); upper lane: the lower lane's gp 1
                            // (16..23), own gp 1 (24..31)
                            const int4 q = kq ? make_int4((int)s0[0], (int)s1[0], w0, w1) : make_int4(q8[0], q8[1], (int)s0[1], (int)s1[1]);
                            if (m < a.M) *(int4*)(a.out8 + (long)m * a.out8_cs + n0 + (wn * NREP + j) * 32 + kq * 16) = q;
                        }
                    }
                    if constexpr (EPI == 0) {
                        if (m < a.M) *(uint4*)((_Float16*)a.out + (long)m * a.out_cs + a.out_co + nl) = o.u;
                    } else {
                        *(uint4*)(smem + stg_base + fr * STG_PITCH + (j * 32 + gp * 16 + kq * 8) * 2) = o.u;
                    }
                }
            }
            if constexpr (EPI == 1) {
                // the wave's 32 x (NREP * 32) block leaves as whole rows: NREP * 4 lanes per pixel
                constexpr int CPP = NREP * 4;   // 16-byte chunks per pixel
                const int mb = m0 + (wm * MREP + i) * 32;
                const int nw0 = n0 + wn * NREP * 32;
#pragma unroll
                for (int it = 0; it < (32 * CPP) / 64; ++it) {
                    const int f = it * 64 + lane;
                    const int px = f / CPP, ch = f % CPP;
                    const uint4 vv = *(const uint4*)(smem + stg_base + px * STG_PITCH + ch * 16);
                    if (mb + px < a.M) *(uint4*)((_Float16*)a.out + (long)(mb + px) * a.out_cs + a.out_co + nw0 + ch * 8) = vv;
                }
            }
        }

        if (!has_next) break;
        vb = vbn;
        m0 = m0n;
        n0 = n0n;
    }
    wait_vm<0>();
}


// f16 NHWC view (pixel pitch cs, first channel co, C channels) -> e4m3 rows of `pitch` bytes, zero beyond C
__global__ __launch_bounds__(256) void quant_f8_kernel(const __half* __restrict__ in, int cs, int co, int C, unsigned char* __restrict__ out,
                                                        int pitch, long npix) {
    const int groups = pitch / 16;   // 16 channels per thread
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npix * groups) return;
    const long p = idx / groups;
    const int g = (int)(idx % groups);
    union {
        uint4 u[2];
        _Float16 h[16];
    } x;
    const int c0 = g * 16;
    if (c0 + 16 <= C) {
        const uint4* src = (const uint4*)(in + p * cs + co + c0);
        x.u[0] = src[0];
        x.u[1] = src[1];
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) x.h[e] = c0 + e < C ? (_Float16)in[p * cs + co + c0 + e] : (_Float16)0.f;
    }
    union {
        uint4 u;
        int w[4];
    } o;
    // the conversion turns what lies beyond e4m3's range into NaN: saturate at +-448 first, as the packer does
    const auto sat = [](_Float16 v) { return __builtin_amdgcn_fmed3f((float)v, -448.f, 448.f); };
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(sat(x.h[4 * e + 0]), sat(x.h[4 * e + 1]), w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(sat(x.h[4 * e + 2]), sat(x.h[4 * e + 3]), w, true);
        o.w[e] = w;
    }
    *(uint4*)(out + p * pitch + c0) = o.u;
}

struct T32F8Tile {
    int bm, bn, threads, a_slots, ring, nrep, epi, wgs_per_cu;
    void (*kernel)(const ConvArgs, int, int, int);
};

#define T32F8(WM, WN, MR, NR, AS, R, EPI, WPC) \
    { WM * MR * 32, WN * NR * 32, WM * WN * 64, AS, R, NR, EPI, WPC, conv_t32f8_kernel<WM, WN, MR, NR, AS, R, EPI, WPC> }

const T32F8Tile kT32F8Tiles[] = {
    T32F8(4, 2, 2, 3, 4, 5, 0, 1),    // 0: 256 x 192, 40-wide maps
    T32F8(4, 2, 2, 3, 4, 4, 0, 1),    // 1: 256 x 192, up to 80-wide maps
    T32F8(8, 1, 2, 3, 10, 5, 0, 1),   // 2: 512 x 96
    T32F8(4, 2, 2, 4, 8, 4, 0, 1),    // 3: 256 x 256 (fused head convs)
    T32F8(8, 1, 2, 2, 12, 5, 0, 1),   // 4: 512 x 64
    T32F8(8, 1, 1, 3, 10, 4, 0, 2),   // 5: 256 x 96, two workgroups per CU
    T32F8(4, 2, 1, 3, 4, 4, 0, 2),    // 6: 128 x 192, two workgroups per CU
    T32F8(8, 1, 1, 2, 12, 4, 0, 2),   // 7: 256 x 64, two workgroups per CU
};
constexpr int kNumT32F8Tiles = sizeof(kT32F8Tiles) / sizeof(kT32F8Tiles[0]);

int f8_rows(int bm, int W) { return (bm + 2 * W + 2 + 15) / 16 * 16; }
int f8_lds_bytes(const T32F8Tile& t, int W, int cout_pad) { return 2 * f8_rows(t.bm, W) * 64 + t.ring * t.bn * 64 + 1024 + cout_pad * 8; }

}  // namespace

int conv_t32f8_num_tiles() { return kNumT32F8Tiles; }
ConvTile conv_t32f8_tile(int id) { return ConvTile{kT32F8Tiles[id].bm, kT32F8Tiles[id].bn, 64}; }

static bool f8_tile_fits(const T32F8Tile& t, int cout_pad, int W) {
    const int na = f8_rows(t.bm, W) / 16;
    return cout_pad % t.bn == 0 && na <= t.a_slots * (11 - t.ring) && f8_lds_bytes(t, W, cout_pad) <= 160 * 1024 / t.wgs_per_cu;
}

int conv_t32f8_first_tile(int cout_pad, int W) {
    for (int i = 0; i < kNumT32F8Tiles; ++i)
        if (f8_tile_fits(kT32F8Tiles[i], cout_pad, W)) return i;
    return -1;
}

bool conv_t32f8_supported(const ConvArgs& a, int tile) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin % 16 || a.Cin < 32) return false;
    if (a.Ho != a.H || a.Wo != a.W || a.pre || a.in_slab_c || a.out_slab_c || !a.wt8 || !a.in8 || !a.wscale) return false;
    if (tile < 0) return true;
    return f8_tile_fits(kT32F8Tiles[tile], a.Cout_pad, a.W);
}

void launch_quant_f8(DeviceCtx& ctx, hipStream_t stream, const __half* in, int cs, int co, int C, unsigned char* out, int pitch, long npix) {
    if (pitch % 16 || pitch < C || cs % 8 || co % 8) fail(RMR_ERR_LOGIC, "quant_f8: misaligned view");
    const long n = npix * (pitch / 16);
    ProfScope ps(ctx.prof, stream, "quant_f8");
    quant_f8_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(in, cs, co, C, out, pitch, npix);
    RMR_HIP(hipGetLastError());
}

void launch_conv_t32f8(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile) {
    if (tile < 0 || tile >= kNumT32F8Tiles) fail(RMR_ERR_INVALID_ARGUMENT, "conv_t32f8: tile %d out of range", tile);
    if (!conv_t32f8_supported(a, tile)) fail(RMR_ERR_LOGIC, "conv_t32f8: layer not supported by tile %d", tile);
    const T32F8Tile& t = kT32F8Tiles[tile];
    if (a.in8_cs % 64 || a.out_cs % 4 || a.out_co % 4) fail(RMR_ERR_LOGIC, "conv_t32f8: misaligned view");
    if (a.out8 && (a.out32 || ((a.out_cs | a.out_co) & 7) || a.out8_cs % 16))
        fail(RMR_ERR_LOGIC, "conv_t32f8: the e4m3 copy of the output needs an f16 output in 8-channel alignment");
    if (a.out8_only && (a.res || !a.out8)) fail(RMR_ERR_LOGIC, "conv_t32f8: an e4m3-only output cannot carry a shortcut (the planner keeps the f16 copy there)");
    if (a.in8_bytes == 0 || a.in8_bytes > 0xf0000000ull || a.wt8_bytes == 0)
        fail(RMR_ERR_LOGIC, "conv_t32f8: buffer sizes not set or input view larger than 3.75 GiB");
    static std::once_flag once;
    std::call_once(once, [] {
        for (const T32F8Tile& d : kT32F8Tiles)
            (void)hipFuncSetAttribute((const void*)d.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int rows = f8_rows(t.bm, a.W);
    const int lds = f8_lds_bytes(t, a.W, a.Cout_pad);
    const int n_tiles = ((a.M + t.bm - 1) / t.bm) * (a.Cout_pad / t.bn);
    const int grid = std::min((n_tiles + 7) / 8 * 8, ctx.num_cus * t.wgs_per_cu);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = (double)a.N * a.H * a.W * a.Cin + 2.0 * (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K;
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f8";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d f%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, tile);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    t.kernel<<<grid, t.threads, lds, stream>>>(a, rows, n_tiles, 0);
    RMR_HIP(hipGetLastError());
}

// ---- e4m3 on the host -----------------------------------------------------------------------------------
// OCP e4m3fn: 1 sign, 4 exponent bits (bias 7), 3 mantissa bits; no infinities, 0x7f / 0xff = NaN, largest
// finite value 448; round to nearest even, values beyond 448 saturate (the weights are scaled so that none is)
unsigned char f32_to_e4m3(float x) {
    if (std::isnan(x)) return 0x7f;
    const unsigned char sign = std::signbit(x) ? 0x80 : 0;
    float a = std::fabs(x);
    if (a >= 464.f) return sign | 0x7e;             // 448 is the largest finite value (464 = the midpoint to 480)
    if (a < 0.0009765625f) return sign;             // below half of the smallest subnormal 2^-9
    int e;
    std::frexp(a, &e);                              // a = m * 2^e, m in [0.5, 1)
    int exp = e - 1;                                // a = 1.f * 2^exp
    if (exp < -6) exp = -6;                         // subnormal range: fixed exponent, no hidden bit
    const float q = std::ldexp(1.0f, exp - 3);      // value of one mantissa step
    float steps = a / q;                            // exact: a power-of-two scaling
    float r = std::nearbyint(steps);                // round half to even (default rounding mode)
    float v = r * q;
    if (v > 448.f) v = 448.f;
    // re-derive exponent and mantissa of the rounded value
    if (v < 0.015625f) return sign | (unsigned char)std::lrint(v / 0.001953125f);   // subnormal: mantissa = v / 2^-9
    int e2;
    const float m2 = std::frexp(v, &e2);            // v = m2 * 2^e2, m2 in [0.5, 1)
    const int be = e2 - 1 + 7;
    const int man = (int)std::lrint((m2 * 2.f - 1.f) * 8.f);
    return sign | (unsigned char)((be << 3) | man);
}

float e4m3_to_f32(unsigned char b) {
    const int be = (b >> 3) & 15, man = b & 7;
    if (be == 15 && man == 7) return std::nanf("");
    const float v = be == 0 ? std::ldexp((float)man, -9) : std::ldexp(1.f + man / 8.f, be - 7);
    return b & 0x80 ? -v : v;
}

// [Cout_pad][Kp] f16 (k = tap * Cin + ci) -> e4m3 with one scale per output channel (the largest |w| of a row
// becomes 448), as the LDS images of the (64-channel chunk, tap) slices:
// [chunk][tap][Cout_pad / 16][64 lanes][16]: lane l = row l >> 2, slot l & 3 holds channels 16 (slot ^ key(row)) ..
void pack_conv_weights_t32f8(const __half* packed, int cout_pad, int cin, int Kp, std::vector<unsigned char>& out,
                             std::vector<float>& scale) {
    const int chunks = (cin + 63) / 64, nblk = cout_pad / 16;
    out.assign((size_t)chunks * 9 * nblk * 1024, 0);
    scale.assign(cout_pad, 1.f);
    for (int n = 0; n < cout_pad; ++n) {
        float mx = 0.f;
        for (int k = 0; k < 9 * cin; ++k) mx = std::max(mx, std::fabs(__half2float(packed[(size_t)n * Kp + k])));
        scale[n] = mx > 0.f ? mx / 448.f : 1.f;
    }
    for (int cc = 0; cc < chunks; ++cc)
        for (int t = 0; t < 9; ++t)
            for (int b = 0; b < nblk; ++b)
                for (int l = 0; l < 64; ++l) {
                    const int r = l >> 2, s = l & 3;
                    const int n = b * 16 + r;
                    const int c = s ^ ((r >> 2) & 3);
                    unsigned char* dst = out.data() + ((((size_t)cc * 9 + t) * nblk + b) * 64 + l) * 16;
                    for (int e = 0; e < 16; ++e) {
                        const int ci = cc * 64 + c * 16 + e;
                        dst[e] = ci < cin ? f32_to_e4m3(__half2float(packed[(size_t)n * Kp + (size_t)t * cin + ci]) / scale[n]) : 0;
                    }
                }
}

}  // namespace rmr
