// conv_t32.hip -- 3x3 / stride 1 / pad 1 convolution on v_mfma_f32_32x32x16_f16, one 8-wave
// workgroup per CU, written so that the K loop carries almost no scalar or vector-ALU work.
//
// conv_halo (round 1) stages the input range once per 32-channel chunk and takes the nine taps as
// row shifts of the fragment reads; measured, its K loop is not short of bytes but of issue slots
// and overlap: 7.6 VALU + 2.7 SALU per 16x16x32 MFMA, one barrier per 12-24 MFMAs with the fragment
// reads AFTER it, 45 % of the wave time parked.  This kernel keeps the halo staging and changes
// everything around it:
//
//   * 32x32x16 MFMAs (half the matrix instructions per FLOP, 32-cycle issue slots to hide the rest
//     in), a wave tile of 64 pixels x 96..128 channels, 8 waves = 256 x 192 or 512 x 96 per CU so that
//     the weight stream is shared by 256-512 pixels (one workgroup per CU, up to 256 VGPRs);
//   * weights are re-packed on the host into the exact LDS image of a (chunk, tap) slice, swizzle
//     included, so a weight DMA instruction reads ONE contiguous KiB (eight whole 128-byte lines)
//     from a wave-uniform offset: lane * 16 in the VGPR, everything else in the scalar offset;
//   * fragment reads run half a tap ahead of the MFMAs that consume them, ACROSS the barrier: a
//     slice is waited for (counted vmcnt) one tap before its first read, so the reads of tap t + 1
//     are legal before the barrier that opens tap t + 1, and the MFMAs behind a barrier start at once;
//   * border taps are masked by redirecting the fragment read to a zero block, as before, but the
//     nine validity bits of a lane live in SGPR pairs as lane masks (one v_cndmask per fragment and
//     tap), and fragment i of a wave sits at a constant 2 KiB from fragment 0 (immediate offsets):
//     the swizzle key has a period of 16 rows, so ONE address is computed per tap.
//
// LDS rows are 64 bytes (32 channels of one pixel / one output channel).  A lane of a 32x32x16
// fragment read (row = lane & 31, k-half = lane >> 5) takes the 16-byte chunk (2 h + k-half) of its
// row for the MFMA of K-step h; chunk c of row r is stored at slot c ^ ((r >> 2) & 3), which makes
// every ds_read_b128 lane group cover all 64 banks once at every row shift.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include "conv_t32_common.h"

namespace rmr {

namespace {

using namespace t32;

// WM x WN waves, MREP x NREP fragments of 32 x 32 per wave, A_SLOTS input-range DMA instructions per
// tap (the first 11 - R taps of a chunk carry the next chunk's range), R weight slices in the ring.
// EPI: 0 = results leave through v_permlane32_swap pairs as 16-byte stores (32 contiguous bytes per pixel),
//      1 = through a per-wave LDS stage as whole rows (NREP * 64 contiguous bytes per pixel).
//
// PERSISTENT: the grid is at most a few workgroups per CU and a workgroup walks tiles vb = b, b + G, ...
// The DMA stream does not stop at a tile's end: the last R - 1 taps of a tile fetch the first weight
// slices of the next one and its last chunk fetches the next tile's first input range, so only the very
// first tile of a workgroup pays a cold start, and the epilogue of a tile runs while the next tile's
// operands are already in flight.
// ABL (development builds only, -DRMR_T32_ABLATE): bit 0 = no MFMAs, 1 = no epilogue, 2 = epilogue without stores,
// 3 = every DMA out of range (zeros arrive, no memory traffic), 4 = no fragment reads in the K loop.  Timing only.
template <int WM, int WN, int MREP, int NREP, int A_SLOTS, int R, int EPI, int ABL = 0>
__global__ __launch_bounds__(WM* WN * 64, (MREP * NREP > 8 ? 1 : WM * WN <= 4 ? 2 : MREP == 1 ? 4 : 2)) void conv_t32_kernel(const ConvArgs a, const int a_rows, const int n_tiles, const int stagger) {
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
    // UNI: every row of DMA slots has ONE role for all waves, known at compile time -- full weight blocks, then (where the
    // weight blocks of a tap are half a row short) a row of HALF blocks (two waves share a KiB, lanes 0..31 each), then input
    // blocks.  The four-wave tiles with A_SLOTS = 4: 256 x 96 is 6 weight blocks = one full row + one row of halves, + one
    // row of input blocks.  Round 5 had row 1 mixed (waves 0-1 weights, waves 2-3 input): every operand of that DMA went
    // through a select on the wave's role (two s_and + s_cselect pairs and a v_cndmask per tap), and row 2's last two slots
    // fetched the next tap's first input blocks a second time.
    constexpr bool UNI = NW <= 4 && A_SLOTS % NW == 0 && (NB % NW == 0 || NB % NW == NW / 2);
    constexpr int WF = NB / NW, WH = NB % NW ? 1 : 0;   // full and half weight rows (UNI)
    static_assert(!UNI || D == WF + WH + A_SLOTS / NW, "rows of the uniform layout");
    static_assert(R >= 4 && R <= 6, "ring depth");
    static_assert((R - 3) * D <= 63, "vmcnt is 6 bits");
    // (Round 3, measured and dropped: (i) the AccVGPR form of the MFMAs -- hipcc selects the ArchVGPR form for a kernel whose
    // register budget is <= 256 unless the function names an AccVGPR; tools/microbench/mfma_issue.hip has the bare AccVGPR
    // form 12 % / 8 % faster at one / two waves per SIMD on random operands -- forced with one `asm("" : "+a"(x))`: the
    // budget splits 128 + 128, the epilogue pays v_accvgpr_read for every value and spills 16-48 registers, and the
    // layers run 2-5 % SLOWER (288 vs 282, 358 vs 340, 148 vs 143 us).  (ii) A ping-pong form: the eight waves as two
    // groups of four (waves w and w + 4 share a SIMD: tools/microbench/wave_simd.hip) that alternate a bare 12-MFMA
    // segment with a filler segment (10 ds_read_b128, the DMA slots, the counted wait) between s_barriers, as the guide's
    // 8-wave attention loop does: bit-exact, and 534 us against 282 on M409600 N192 K1728 -- its MFMA segments alone take
    // 307 us, its filler segments alone 191, together 534: next to a wave that streams bare MFMAs the partner's
    // vector-memory and LDS instructions do not get issued, so the segments add up instead of overlapping.)

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
    const int bias_off = zero_off + 1024 + (EPI == 1 ? NW * 32 * STG_PITCH : 0);   // Cout_pad floats

    // tile vb -> (m0, n0); XCD-aware: the tiles of one XCD (vb & 7) are a contiguous range, n-tiles innermost
    const int nt_count = a.Cout_pad / BN;
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
    const int G = gridDim.x;  // a multiple of 8: vb & 7 is this workgroup's XCD for every tile it walks
    // split-K (a.split > 1; small batches, where the tiles alone cannot fill the chip): `split` consecutive logical ids
    // (same XCD) share one output tile, each accumulating a contiguous range of 32-channel chunks; the launch has one
    // workgroup per (tile, split), so nobody walks on to a second tile
    const int split = a.split > 1 ? a.split : 1;
    int zsplit = 0, tile_id = 0;
    // The walk (vb = this workgroup's k-th tile): the tiles of an XCD are a contiguous range of logical ids; a workgroup takes
    // every (G / 8)-th tile of that range.  (Round 3 also tried CONTIGUOUS blocks per workgroup, so that a workgroup's next
    // tile reads rows its XCD's L2 already holds: -2 % / +1 % on the two main layers, -1 % frames/s -- neighbouring tiles
    // already run side by side on one XCD and share the L2 fill; DESIGN.md "Round 3".  Removed in round 4.)
    const int my_xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, Gx = G >> 3;
    const int cnt_x = q8 + (my_xcd < r8 ? 1 : 0);
    const int base_x = my_xcd < r8 ? my_xcd * (q8 + 1) : r8 * (q8 + 1) + (my_xcd - r8) * q8;
    const auto tile_valid = [&](int k) { return jx + k * Gx < cnt_x; };
    const auto tile_m0n0 = [&](int k, int& m0, int& n0) {
        const int lid = base_x + jx + k * Gx;
        tile_id = lid / split;
        zsplit = lid - tile_id * split;
        m0 = (tile_id / nt_count) * BM;
        n0 = (tile_id % nt_count) * BN;
    };
    int vb = 0;
    if (!tile_valid(0)) return;
#ifdef RMR_T32_FINISH
    // development build: when every workgroup started and finished (100 MHz clock), for the spread of the static walk
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
#endif
    int m0, n0;
    tile_m0n0(vb, m0, n0);
    const int my_tile = tile_id, my_z = zsplit;
    const int W = a.W;
    const int npix = a.M;        // stride 1: input and output pixels share the linear index

    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu),
                           sgpr((ABL & (8 | 64)) ? 0u : a.in_bytes), sgpr(0x00020000u)};
    const u32x4 wt_rsrc = {sgpr((unsigned)(size_t)a.wt_t32), sgpr((unsigned)((size_t)a.wt_t32 >> 32) & 0xffffu),
                           sgpr((ABL & (8 | 32)) ? 0u : a.wt_t32_bytes), sgpr(0x00020000u)};

    // ---- DMA constants of this lane ----------------------------------------------------------
    const int lrow = lane >> 2;                                   // row inside a 16-row DMA block
    const int lch = (lane & 3) ^ ((lrow >> 2) & 3);               // logical 16-byte chunk it fetches
    const unsigned cs2 = (unsigned)a.in_cs * 2u;
    const unsigned in_cb = (unsigned)((a.in_co + lch * 8) * 2);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int na = a_rows / 16;                                   // input-range DMA blocks per chunk
    const int chunks = a.Cin / 32;
    const int cc_begin = chunks * my_z / split, cc_end = chunks * (my_z + 1) / split;   // this workgroup's chunks
    const int g0 = cc_begin * 9;       // its first weight slice
    const int total = cc_end * 9;      // one past its last
    const unsigned wstep = (unsigned)(a.Cout_pad / 16) * 1024u;   // bytes of one (chunk, tap) slice of all channels
    const unsigned scratch = sgpr(lds0 + zero_off);

    if (tid < 4) *(u32x4*)(smem + zero_off + tid * 16) = u32x4{0, 0, 0, 0};
    for (int i = tid; i < a.Cout_pad; i += NW * 64) *(float*)(smem + bias_off + i * 4) = a.bias[i];

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
        // D * NW can exceed SLOTS (the four-wave tiles: 12 slots for 10 blocks per tap): a surplus slot fetches nothing.  (Until
        // round 6 it fetched block t * A_SLOTS + A_SLOTS (+ 1) -- the next tap's first blocks, a second time: 12 of a chunk's
        // 93 KiB of DMA on the 256 x 96 tile.)  An index past every range keeps it dead in every tap.
        s_aidx[j] = q < SLOTS ? q - NB : 1 << 20;
    }
    // UNI: the lane offset of a weight row carries the row's block (and half) offset, so the scalar offset of every weight
    // DMA of a tap is the stream position itself
    unsigned wvq[D];
#pragma unroll
    for (int j = 0; j < D; ++j) wvq[j] = lane16;
    if constexpr (UNI) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (j < WF) {
                s_wsrc[j] = (unsigned)(wave + NW * j) * 1024u;
            } else if (j < WF + WH) {
                s_wsrc[j] = (unsigned)(NW * WF + (wave >> 1)) * 1024u + (unsigned)(wave & 1) * 512u;
            } else {
                s_wsrc[j] = 0u;
                s_aidx[j] = wave + NW * (j - WF - WH);
            }
            s_wdst[j] = lds0 + ring_base + s_wsrc[j];
            wvq[j] = lane16 + s_wsrc[j];
        }
    }

    // Two workgroups share a CU so that one's epilogue (two transcendentals per output value: a third of the
    // MFMA time of a K = 864 tile) runs under the other's K loop -- which only happens when they are out of
    // phase.  Launched together and walking equal tiles they would stay in lockstep for the whole launch, so the
    // workgroup in the CU's odd slot (HW_ID.TG_ID, the barrier resource it was given) starts half a tile late.
    // prio (RMR_T32_PRIO, default 1): 1 = the K loop runs at s_setprio 1 and the epilogue at 0 -- the other workgroup of the
    // CU keeps the matrix pipe fed while this one does its SiLUs and stores (isolated: M1638400 N96 K864 346 -> 331 us,
    // M409600 N192 K1728 282 -> 279; in the bench 1945-1947 -> 1952-1963 frames/s); 2 = the reverse (no gain); 0 = off
    const int prio = (stagger >> 16) & 15;
    if ((stagger & 0xffff) > 0) {
        const unsigned hw_id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 16 << 6 | 4);   // HW_REG_HW_ID[19:16] = TG_ID
        if (hw_id & 1)
            for (int i = 0; i < (stagger & 0xffff); ++i) __builtin_amdgcn_s_sleep(64);   // 64 x 64 cycles each
    }

    const auto dma_in = [](u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
        if constexpr (ABL & 512)
            dma16s_nt(rsrc, lds_addr, voff, soff);
        else
            dma16s(rsrc, lds_addr, voff, soff);
    };
    // ---- cold start: the whole input range of chunk 0, weight slices 0 .. R-2 of the first tile -----------
    {
        const int pl0 = m0 - W - 1 + lrow;
        for (int ia = wave; ia < na; ia += NW) dma_in(in_rsrc, sgpr(lds0 + ia * 1024), in_off(pl0, ia, cc_begin), 0u);
#pragma unroll
        for (int s = 0; s < R - 1; ++s)
            for (int q = wave; q < NB; q += NW)
                dma16s(wt_rsrc, sgpr(lds0 + ring_base + s * SLOT_BYTES + q * 1024), g0 + s < total ? lane16 : OOB,
                       sgpr((unsigned)(g0 + s) * wstep + (unsigned)(n0 / 16 + q) * 1024u));
    }

    // ---- fragment constants ------------------------------------------------------------------------
    const int fr = lane & 31, kq = lane >> 5;
    const int a_row0 = wm * MREP * 32 + fr + W + 1;   // LDS row of the centre tap of fragment 0
    int zsel[MREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i) zsel[i] = zero_off - i * 2048;
    const int wlane = ring_base + (wn * NREP * 32 + fr) * 64 + ((kq ^ ((fr >> 2) & 3)) << 4);
    const auto lds16 = [&](int off) {
        if constexpr (ABL & 16) {
            half8 z = {};
            asm volatile("" : "+v"(z) : "v"(off));
            return z;
        } else {
            return *(const half8*)(smem + off);
        }
    };
    const auto mma = [](half8 w, half8 x, floatx16 c) {
        if constexpr (ABL & 1) {
            asm volatile("" : "+v"(c) : "v"(w), "v"(x));
            return c;
        } else {
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, c, 0, 0, 0);
        }
    };
    // address of fragment 0's chunk for K-step 0 of tap t, in input buffer `abuf`
    const auto a_addr = [&](int abuf, int t) {
        const int row = a_row0 + (t / 3 - 1) * W + (t % 3 - 1);
        return abuf + row * 64 + ((kq ^ ((row >> 2) & 3)) << 4);
    };

    wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    constexpr int NM = MREP * NREP;   // MFMAs per K-step
    // Scalar state of the K loop, kept to a handful of SGPRs with one or two SALU instructions per tap.  (Round 5's form --
    // slot numbers multiplied out per tap, a slice counter compared with the tile's total in every tap, four selects behind
    // it -- was 34 SALU instructions per tap, and the compiler computed all nine taps' worth in FRONT of a chunk's first MFMA
    // and parked the results in VGPR lanes: 196 spilled SGPRs, 113 v_readlane + 28 v_writelane + 59 s_nop per chunk.)
    int woff = 0;                               // ring slot of the tap being computed, as a byte offset
    int woff_prev = (R - 1) * SLOT_BYTES;       // the slot the previous tap left: where this tap's weight DMAs land
    int abuf = 0;                               // input buffer of the chunk being computed
    // the weight stream: the next slice to fetch lies at byte offset wsoff (channel-tile offset + slice offset) of the packed
    // weights; the stream runs R - 1 slices ahead of the MFMAs, so it crosses into the next tile's weights behind tap 9 - R of
    // a tile's LAST chunk -- a compile-time tap, not a counter
    unsigned wsoff = (unsigned)(n0 / 16) * 1024u + (unsigned)(g0 + R - 1) * wstep;
    unsigned wv = lane16;                       // the weight DMAs' lane offset: OOB once the stream has run past the last tile

    for (;;) {
        if (prio == 1) __builtin_amdgcn_s_setprio(1);
        if (prio == 2) __builtin_amdgcn_s_setprio(0);
        // ---- this tile and the next one -------------------------------------------------------------
        const int vbn = vb + 1;
        const bool has_next = tile_valid(vbn);
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

        // fragments of (tap 0, K-step 0): the slice and the range were waited for before the last barrier
        half8 xa[MREP], wa[NREP], xb[MREP], wb[NREP];
        int selx[MREP];
        int wcur = wlane + woff;
        {
            const int at = a_addr(abuf, 0);
#pragma unroll
            for (int i = 0; i < MREP; ++i) {
                selx[i] = (up[i] && lf[i]) ? at : zsel[i];
                xa[i] = lds16(selx[i] + i * 2048);
            }
#pragma unroll
            for (int j = 0; j < NREP; ++j) wa[j] = lds16(wcur + j * 2048);
        }

        for (int cc = cc_begin; cc < cc_end; ++cc) {
            const int abuf_next = a_buf_bytes - abuf;
            // the range fetched during this chunk: the next chunk of this tile, or chunk 0 of the next tile
            const bool in_tile = cc + 1 < cc_end;
            const bool a_live = in_tile || has_next;
            const int a_pl = in_tile ? pl : pln;
            const int a_cc = in_tile ? cc + 1 : 0;
            const int na_live = a_live ? na : 0;                       // blocks of the range fetched during this chunk
            const unsigned a_base = lds0 + (unsigned)abuf_next;        // where they land
            // One tap = 2 NM MFMAs (K-step 0, then K-step 1); everything else is placed by hand into the gaps
            // behind them (a 32x32x16 MFMA occupies the pipe for 32 cycles).  The fragments of a K-step are read
            // one K-step ahead: those of the next tap's K-step 0 BEFORE the barrier that opens that tap, so the
            // first MFMAs behind a barrier never wait for the LDS.
            const auto tap = [&](auto T) {
                constexpr int t = decltype(T)::value;
                constexpr int tn = (t + 1) % 9;
                int at_n = 0;
                __builtin_amdgcn_s_barrier();
                // fillers of K-step 0: the K-step 1 fragments of this tap (pixels, then weights), D DMA slots, the
                // next tap's addresses; filler f rides behind MFMA f * NM / (D + 3)
                const auto filler0 = [&](auto Fc) {
                    constexpr int f = decltype(Fc)::value;
                    if constexpr (f == 0) {
#pragma unroll
                        for (int i = 0; i < MREP; ++i) xb[i] = lds16((selx[i] ^ 32) + i * 2048);
                    } else if constexpr (f == 1) {
#pragma unroll
                        for (int j = 0; j < NREP; ++j) wb[j] = lds16((wcur ^ 32) + j * 2048);
                    } else if constexpr (f < 2 + D) {
                        // ---- DMA slot d: the next weight slice of the stream into the ring slot tap g - 1 left,
                        // or one block of the next input range
                        constexpr int d = f - 2;
                        constexpr bool all_w = NW * (d + 1) <= NB, all_a = NW * d >= NB;
                        constexpr bool a_tap = t < ATAPS;
                        if constexpr (UNI) {
                            if constexpr (d < WF) {
                                dma16s(wt_rsrc, sgpr(s_wdst[d] + woff_prev), wvq[d], sgpr(wsoff));
                            } else if constexpr (d < WF + WH) {
                                dma16s_lo(wt_rsrc, sgpr(s_wdst[d] + woff_prev), wvq[d], sgpr(wsoff));
                            } else if constexpr (a_tap) {
                                const int ia = t * A_SLOTS + s_aidx[d];
                                const bool alive = ia < na_live;
                                unsigned av = in_off(a_pl, ia, a_cc);
                                asm volatile("" : "+v"(av));   // computed unconditionally: a branch around it would split the tap's basic block
                                dma_in(in_rsrc, sgpr(alive ? a_base + ia * 1024 : scratch), alive ? av : OOB, 0u);
                            }
                        } else {
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
                                    dma_in(in_rsrc, sgpr(a_lds), alive ? av : OOB, 0u);
                                }
                        } else {
                            const bool isw = s_isw[d];
                            unsigned av = in_off(a_pl, ia, a_cc);
                                asm volatile("" : "+v"(av));   // computed unconditionally: a branch around it would split the tap's basic block
                                av = (a_tap && alive) ? av : OOB;
                            dma16s(s_rsrc[d], sgpr(isw ? w_lds : a_lds), isw ? wv : av, sgpr(isw ? w_soff : 0u));
                        }
                        }
                    } else {
                        at_n = a_addr(t == 8 ? abuf_next : abuf, tn);
                        woff_prev = woff;
                        woff = woff + SLOT_BYTES == R * SLOT_BYTES ? 0 : woff + SLOT_BYTES;
                        wcur = wlane + woff;
                    }
                };
                static_for<0, NM>([&](auto Kc) {
                    constexpr int k = decltype(Kc)::value;
                    acc[k / NREP][k % NREP] = mma(wa[k % NREP], xa[k / NREP], acc[k / NREP][k % NREP]);
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, D + 3>([&](auto Fc) {
                        if constexpr (decltype(Fc)::value * NM / (D + 3) == k) filler0(Fc);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                static_for<0, NM>([&](auto Kc) {
                    constexpr int k = decltype(Kc)::value;
                    acc[k / NREP][k % NREP] = mma(wb[k % NREP], xb[k / NREP], acc[k / NREP][k % NREP]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (k == 0) {
                        // K-step 0 fragments of the next tap (tap 0 of the next chunk after tap 8): legal before the
                        // next barrier because that slice was waited for one tap ago.  (After the tile's last tap
                        // they are read in vain: the next tile's lane masks are not known here.)
                        constexpr int dy = tn / 3 - 1, dx = tn % 3 - 1;
#pragma unroll
                        for (int i = 0; i < MREP; ++i) {
                            const bool v = (dy < 0 ? up[i] : dy > 0 ? dn[i] : true) && (dx < 0 ? lf[i] : dx > 0 ? rt[i] : true);
                            selx[i] = v ? at_n : zsel[i];
                            xa[i] = lds16(selx[i] + i * 2048);
                        }
                    }
                    if constexpr (k == (NM > 1 ? 1 : 0)) {
#pragma unroll
                        for (int j = 0; j < NREP; ++j) wa[j] = lds16(wcur + j * 2048);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // DMAs issued R - 3 taps ago (and earlier) have landed -- and, behind an epilogue, its stores, which
                // are older than this tap's DMAs (loads and stores retire in order)
                constexpr int pending = [] {
                    int n = 0;
                    for (int k = 0; k < R - 3; ++k) {
                        const int tt = (t - k + 9) % 9;
                        for (int j = 0; j < D; ++j) n += (NW * j >= NB && tt >= ATAPS) ? 0 : 1;
                    }
                    return n;
                }();
                if constexpr (!(ABL & 128)) wait_vm<pending>();
                // advance the weight stream; behind a tile's last slice comes the first one of the next tile
                // (selects, not a branch: a tap must stay one basic block, or the scheduling pins above do not hold
                // the MFMAs in place and the compiler sinks them towards the end of the chunk)
                if constexpr (t == 9 - R) {
                    wsoff = in_tile ? wsoff + wstep : w_tile_next;
                    if constexpr (UNI) {
#pragma unroll
                        for (int j = 0; j < WF + WH; ++j) wvq[j] = in_tile ? wvq[j] : (has_next ? lane16 + s_wsrc[j] : OOB);
                    } else {
                        wv = in_tile ? wv : (has_next ? lane16 : OOB);
                    }
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

        // ---- split-K: the partial tiles meet in the workspace; the last arriver sums ALL of them in split order (its own
        // re-read from the workspace), so the result does not depend on who arrived last: run-to-run deterministic
        if (split > 1) {
            constexpr int TILE_F4 = NW * MREP * NREP * 64 * 4;  // float4 per partial tile
            constexpr int SC = 17;                              // sc0 sc1: write-through stores, loads that bypass L1 and L2
            int* const is_last = (int*)(smem + bias_off + a.Cout_pad * 4);
            // The partial tiles travel as write-through (sc0 sc1) 16-byte stores and are read back with sc0 sc1 loads: no
            // agent-scope release / acquire fence (an L2 write-back and an L1 invalidate: 5-13 us per seam on this chip, as
            // much as the K loop of a batch-1 layer), only the wave's own vmcnt(0) before the ticket (guide, "publish-large")
            const __amdgpu_buffer_rsrc_t ws_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((float4*)a.splitk_ws + (size_t)my_tile * split * TILE_F4), 0, (unsigned)((size_t)split * TILE_F4 * 16), 0x00020000);
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const u32x4 v = {__float_as_uint(acc[i][j][4 * r]), __float_as_uint(acc[i][j][4 * r + 1]), __float_as_uint(acc[i][j][4 * r + 2]),
                                         __float_as_uint(acc[i][j][4 * r + 3])};
                        __builtin_amdgcn_raw_buffer_store_b128(v, ws_rsrc, (unsigned)((my_z * TILE_F4 + (((wave * MREP + i) * NREP + j) * 4 + r) * 64 + lane) * 16), 0, SC);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const int ticket = __hip_atomic_fetch_add(a.splitk_cnt + my_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *is_last = ticket == split - 1;
                if (ticket == split - 1) __hip_atomic_store(a.splitk_cnt + my_tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
            }
            __syncthreads();
            if (!*is_last) return;
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            for (int z = 0; z < split; ++z) {
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < NREP; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const u32x4 p = __builtin_amdgcn_raw_buffer_load_b128(ws_rsrc, (unsigned)((z * TILE_F4 + (((wave * MREP + i) * NREP + j) * 4 + r) * 64 + lane) * 16), 0, SC);
                            acc[i][j][4 * r] += __uint_as_float(p[0]), acc[i][j][4 * r + 1] += __uint_as_float(p[1]);
                            acc[i][j][4 * r + 2] += __uint_as_float(p[2]), acc[i][j][4 * r + 3] += __uint_as_float(p[3]);
                        }
            }
        }

        // ---- epilogue; the next tile's first slices and input range are in flight meanwhile
        if (prio == 1) __builtin_amdgcn_s_setprio(0);
        if (prio == 2) __builtin_amdgcn_s_setprio(1);
        if constexpr (ABL & 2) {
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" : : "v"(acc[i][j]));
#endif
                }
        } else {
            // EPI = 2: rows staged through the input buffer the tile's last chunk has just finished with (the other one is
            // receiving the next tile's first range; this one is not written again before every wave has passed the next
            // tile's first barrier): whole-row stores for the tiles that have no LDS to spare for a stage of their own
            if constexpr (EPI == 2) __builtin_amdgcn_s_barrier();   // every wave has read its last fragments from that buffer
            const int stg = EPI == 2 ? (a_buf_bytes - abuf) + wave * 32 * STG_PITCH : stg_base;
            epilogue<MREP, NREP, (EPI == 2 ? 1 : EPI), true, (NREP < 4 && MREP * NREP <= 8), (ABL & 4) != 0, (ABL & 256) ? 2 : 0>(a, acc, smem, stg, bias_off, m0, n0, wm, wn, lane);
        }

        if (!has_next) break;
        vb = vbn;
        m0 = m0n;
        n0 = n0n;
    }
    wait_vm<0>();
#ifdef RMR_T32_FINISH
    if (split == 1 && a.splitk_ws && threadIdx.x == 0) {
        unsigned long long* const rec = (unsigned long long*)a.splitk_ws + 2 * blockIdx.x;
        rec[0] = t_begin;
        rec[1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}



#ifdef RMR_T32_PINGPONG
#include "../../tools/experiments/conv_t32_pingpong.inc"
#endif

struct T32Tile {
    int bm, bn, threads, a_slots, ring, nrep, epi, wgs_per_cu;
    void (*kernel)(const ConvArgs, int, int, int);
};

#define T32(WM, WN, MR, NR, AS, R, EPI, WPC) \
    { WM * MR * 32, WN * NR * 32, WM * WN * 64, AS, R, NR, EPI, WPC, conv_t32_kernel<WM, WN, MR, NR, AS, R, EPI> }
#define T32A(WM, WN, MR, NR, AS, R, EPI, WPC, ABL) \
    { WM * MR * 32, WN * NR * 32, WM * WN * 64, AS, R, NR, EPI, WPC, conv_t32_kernel<WM, WN, MR, NR, AS, R, EPI, ABL> }

const T32Tile kT32Tiles[] = {
    // one workgroup per CU (up to 256 VGPRs)
    T32(4, 2, 2, 3, 4, 5, 1, 1),    // 0: 256 x 192, 40-wide maps (22 input blocks over 6 taps), rows leave through LDS
    T32(4, 2, 2, 3, 4, 4, 1, 1),    // 1: 256 x 192, up to 80-wide maps (27 blocks over 7 taps)
    T32(4, 2, 2, 3, 4, 5, 0, 1),    // 2: as 0 with lane-pair stores
    T32(8, 1, 2, 3, 10, 5, 0, 1),   // 3: 512 x 96
    T32(4, 2, 2, 4, 8, 4, 0, 1),    // 4: 256 x 256 (fused head convs)
    T32(8, 1, 2, 2, 12, 5, 0, 1),   // 5: 512 x 64
    // two workgroups per CU (<= 128 VGPRs, <= 80 KiB of LDS): one's epilogue under the other's K loop
    T32(8, 1, 1, 3, 10, 4, 0, 2),   // 6: 256 x 96
    T32(8, 1, 1, 2, 12, 4, 0, 2),   // 7: 256 x 64
    T32(4, 2, 1, 3, 4, 4, 0, 2),    // 8: 128 x 192
    // two FOUR-wave workgroups per CU (one wave per SIMD each, full 64 x 96 wave tiles, up to 256 VGPRs): the two
    // drift apart on their own, so one's epilogue runs under the other's K loop, and a barrier joins four waves
    T32(2, 2, 2, 3, 2, 4, 0, 2),    // 9: 128 x 192
    T32(4, 1, 2, 3, 4, 4, 0, 2),    // 10: 256 x 96
    T32(2, 2, 2, 2, 2, 4, 0, 2),    // 11: 128 x 128
    T32(4, 1, 2, 2, 4, 4, 0, 2),    // 12: 256 x 64
    // (tried: ONE four-wave workgroup per CU with a 128 x 96 wave tile, T32(2, 2, 4, 3, 4, 5, 0, 1) -- 192 accumulators, 7
    // fragment reads per 12 MFMAs instead of 10, one wave per SIMD: 304.4 us against 303.1 on M409600 N192 K1728.  Half the
    // waves, 30 % fewer LDS reads, the same time: the K loop sits at the package power limit, not at a pipe.)
    // whole-row stores for the two-per-CU tiles: rows staged through the input buffer the last chunk is done with
    T32(4, 1, 2, 3, 4, 4, 2, 2),    // 13: tile 10 (256 x 96), where that buffer holds the wave stages (80-wide maps)
    T32(2, 2, 2, 3, 2, 4, 2, 2),    // 14: tile 9 (128 x 192)
    // THREE two-wave workgroups per CU (round 6): 128 x 96 with the full 64 x 96 wave tile, for launches
    // whose 256-row tiles are fewer than the chip's workgroup slots (M25600 N288 at 64 images: 300 tiles of 256 x 96 on 512
    // slots).  20-wide maps only: on 40-wide ones the input ranges leave room for two per CU.
    T32(2, 1, 2, 3, 2, 4, 0, 3),    // 15: 128 x 96
    // (tried: tile 10 with a ring of FIVE slices, T32(4, 1, 2, 3, 4, 5, 0, 2) -- one more tap of lookahead for every DMA, fits two
    // per CU up to 40-wide maps: +0 ... 2 % on M409600 N192 K1728 and M102400 N288 K2592, inside the spread of two runs:
    // profiles/r06_ring5.txt.  Not kept: it splits the dominant instantiation's layers over two symbols for nothing.)
#ifdef RMR_T32_PINGPONG
    // 15.. (development builds): the ping-pong form
    { 256, 192, 512, 4, 5, 3, 0, 1, conv_t32pp_kernel<4, 2, 2, 3, 4, 5, 0> },     // 13: 256 x 192
    { 256, 192, 512, 4, 4, 3, 0, 1, conv_t32pp_kernel<4, 2, 2, 3, 4, 4, 0> },     // 14: up to 80-wide maps
    { 512, 96, 512, 10, 5, 3, 0, 1, conv_t32pp_kernel<8, 1, 2, 3, 10, 5, 0> },    // 15: 512 x 96
    { 512, 96, 512, 10, 4, 3, 0, 1, conv_t32pp_kernel<8, 1, 2, 3, 10, 4, 0> },    // 16
#endif
#ifdef RMR_T32_ABLATE
    // 13..25: tile 10 (256 x 96, two four-wave workgroups per CU) with parts removed; 26..: tile 1 (256 x 192)
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 1),    // 13: no MFMA
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 2),    // 14: no epilogue
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 4),    // 15: epilogue without stores
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 8),    // 16: DMAs out of range
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 16),   // 17: no fragment reads
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 10),   // 18: no epilogue, DMAs out of range
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 26),   // 19: MFMAs + barriers only
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 12),   // 20: epilogue without stores, DMAs out of range
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 32),   // 21: weight DMAs out of range
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 64),   // 22: input DMAs out of range
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 128),  // 23: no vmcnt waits in the taps
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 132),  // 24: no waits, no stores
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 130),  // 25: no waits, no epilogue
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 256),  // 26: non-temporal stores
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 512),  // 27: non-temporal input DMAs
    T32A(4, 1, 2, 3, 4, 4, 0, 2, 768),  // 28: both
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 256),  // 29 (256 x 192): non-temporal stores
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 768),  // 30 (256 x 192): both
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 32),   // 31 (256 x 192)
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 64),   // 27
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 128),  // 28
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 132),  // 29
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 1),    // 30
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 2),    // 22
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 4),    // 23
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 8),    // 24
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 16),   // 25
    T32A(4, 2, 2, 3, 4, 4, 1, 1, 26),   // 26
    // tile 3 (512 x 96, eight waves, ONE workgroup per CU: the weight slices are fetched once per 512 pixels)
    T32A(8, 1, 2, 3, 10, 5, 0, 1, 2),   // no epilogue
    T32A(8, 1, 2, 3, 10, 5, 0, 1, 8),   // DMAs out of range
    T32A(8, 1, 2, 3, 10, 5, 0, 1, 10),  // neither
    T32A(8, 1, 2, 3, 10, 5, 0, 1, 32),  // weight DMAs out of range
    T32A(8, 1, 2, 3, 10, 5, 0, 1, 64),  // input DMAs out of range
    T32A(8, 1, 2, 3, 10, 5, 0, 1, 26),  // MFMAs + barriers only
#endif
};
constexpr int kNumT32Tiles = sizeof(kT32Tiles) / sizeof(kT32Tiles[0]);

int t32_rows(int bm, int W) { return (bm + 2 * W + 2 + 15) / 16 * 16; }
int t32_lds_bytes(const T32Tile& t, int W, int cout_pad) {
    return 2 * t32_rows(t.bm, W) * 64 + t.ring * t.bn * 64 + 1024 + (t.epi == 1 ? (t.threads / 64) * 32 * (t.nrep * 64 + 16) : 0) + cout_pad * 4 + 16;
}

}  // namespace

int conv_t32_num_tiles() { return kNumT32Tiles; }
ConvTile conv_t32_tile(int id) { return ConvTile{kT32Tiles[id].bm, kT32Tiles[id].bn, 32}; }

bool conv_t32_supported(const ConvArgs& a, int tile) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin % 32 || a.Cin < 32) return false;
    if (a.Ho != a.H || a.Wo != a.W || a.pre || a.in_slab_c || a.out_slab_c || !a.wt_t32) return false;
    if (tile < 0) return true;
    const T32Tile& t = kT32Tiles[tile];
    const int na = t32_rows(t.bm, a.W) / 16;
    if (t.epi == 2 && (t.threads / 64) * 32 * (t.nrep * 64 + 16) > t32_rows(t.bm, a.W) * 64) return false;   // the stage lives in one input buffer
    return a.Cout_pad % t.bn == 0 && na <= t.a_slots * (11 - t.ring) && t32_lds_bytes(t, a.W, a.Cout_pad) <= 160 * 1024 / t.wgs_per_cu;
}

// split-K: tiles of `tile` on this layer, and the workspace floats `split` partial tiles of each need
int conv_t32_splitk_tiles(const ConvArgs& a, int tile) {
    const T32Tile& t = kT32Tiles[tile];
    return ((a.M + t.bm - 1) / t.bm) * (a.Cout_pad / t.bn);
}
size_t conv_t32_splitk_ws_floats(const ConvArgs& a, int tile, int split) {
    const T32Tile& t = kT32Tiles[tile];
    return (size_t)conv_t32_splitk_tiles(a, tile) * split * t.bm * t.bn;
}
bool conv_t32_splitk_supported(const ConvArgs& a, int tile, int split, int num_cus) {
    if (!conv_t32_supported(a, tile) || split < 2 || split > a.Cin / 32) return false;
    const T32Tile& t = kT32Tiles[tile];
    return !t.epi && (long)conv_t32_splitk_tiles(a, tile) * split <= (long)num_cus * t.wgs_per_cu;
}

void launch_conv_t32(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile) {
    if (tile < 0 || tile >= kNumT32Tiles) fail(RMR_ERR_INVALID_ARGUMENT, "conv_t32: tile %d out of range", tile);
    if (!conv_t32_supported(a, tile)) fail(RMR_ERR_LOGIC, "conv_t32: layer not supported by tile %d", tile);
    const T32Tile& t = kT32Tiles[tile];
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 4 || a.out_co % 4) fail(RMR_ERR_LOGIC, "conv_t32: misaligned view");
    if (a.in_bytes == 0 || a.in_bytes > 0xf0000000ull || a.wt_t32_bytes == 0)
        fail(RMR_ERR_LOGIC, "conv_t32: buffer sizes not set or input view larger than 3.75 GiB");
    static std::once_flag once;
    std::call_once(once, [] {
        for (const T32Tile& d : kT32Tiles)
            (void)hipFuncSetAttribute((const void*)d.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int rows = t32_rows(t.bm, a.W);
    const int lds = t32_lds_bytes(t, a.W, a.Cout_pad);
    int n_tiles = ((a.M + t.bm - 1) / t.bm) * (a.Cout_pad / t.bn);
    if (a.split > 1) {
        if (a.split > a.Cin / 32) a.split = a.Cin / 32;
        if (!a.splitk_ws || !a.splitk_cnt) fail(RMR_ERR_LOGIC, "conv_t32: split-K needs a workspace");
        if (t.epi) fail(RMR_ERR_LOGIC, "conv_t32: split-K runs on the lane-pair-store tiles");
        if ((long)n_tiles * a.split > (long)ctx.num_cus * t.wgs_per_cu) fail(RMR_ERR_LOGIC, "conv_t32: split-K needs one workgroup slot per (tile, split)");
        if (a.split > 1) n_tiles *= a.split;
    }
    // persistent: at most wgs_per_cu workgroups per CU (a multiple of 8: a workgroup stays on its XCD), each walks tiles
    const int grid = std::min((n_tiles + 7) / 8 * 8, ctx.num_cus * t.wgs_per_cu);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        if (a.split > 1)
            snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d g%d/%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, tile, a.split);
        else
            snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d g%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, tile);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    // half a tile of head start for one workgroup of each CU pair: ~500 cycles per tap, in units of 4096 cycles
    static const int stagger_env = std::getenv("RMR_T32_STAGGER") ? std::atoi(std::getenv("RMR_T32_STAGGER")) : -1;
    const int taps = a.Cin / 32 * 9;
    const int stagger = t.wgs_per_cu < 2 || grid <= ctx.num_cus ? 0 : stagger_env >= 0 ? stagger_env : (taps * 500 + 4095) / 4096;
    static const int prio_env = std::getenv("RMR_T32_PRIO") ? std::atoi(std::getenv("RMR_T32_PRIO")) : 1;
    t.kernel<<<grid, t.threads, lds, stream>>>(a, rows, n_tiles, stagger | ((prio_env & 15) << 16));
    RMR_HIP(hipGetLastError());
}

// [Cout_pad][Kp] (k = tap * Cin + ci) -> [chunk][tap][Cout_pad / 16][64 lanes][8]: the LDS image of every
// (chunk, tap) slice in the order the DMA writes it (lane l: row l >> 2, slot l & 3 holds chunk slot ^ key(row))
void pack_conv_weights_t32(const __half* packed, int cout_pad, int cin, int Kp, std::vector<__half>& out, int taps) {
    const int chunks = cin / 32, nblk = cout_pad / 16;
    out.assign((size_t)chunks * taps * nblk * 512, __float2half(0.f));
    for (int cc = 0; cc < chunks; ++cc)
        for (int t = 0; t < taps; ++t)
            for (int b = 0; b < nblk; ++b)
                for (int l = 0; l < 64; ++l) {
                    const int r = l >> 2, s = l & 3;
                    const int n = b * 16 + r;
                    const int c = s ^ ((r >> 2) & 3);
                    const __half* src = packed + (size_t)n * Kp + (size_t)t * cin + cc * 32 + c * 8;
                    __half* dst = out.data() + ((((size_t)cc * taps + t) * nblk + b) * 64 + l) * 8;
                    for (int e = 0; e < 8; ++e) dst[e] = src[e];
                }
}

}  // namespace rmr
