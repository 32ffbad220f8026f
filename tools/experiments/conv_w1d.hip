// conv_w1d.hip -- 3x3 / stride 1 / pad 1 convolution through Winograd F(2, 3) along x, on the conv_t32 skeleton.
//
// The 3x3 K loops of conv_t32 run at the package power limit (DESIGN.md "The power wall"): what is left to gain there is
// energy per output, i.e. fewer matrix instructions.  F(2, 3) computes two neighbouring outputs of a row from a 4-pixel
// window with 4 multiplications per (input channel, filter row) instead of 6:
//
//     V = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]          (input transform, per channel: one f16 add each)
//     U = [g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2]   (filter transform, per (ky, cin, cout): host, f32 -> f16 once)
//     M_xi = sum over (ky, cin) of U[ky][xi] * V_xi(y + ky - 1)        (four GEMMs, the filter rows stay direct)
//     out(2t) = M0 + M1 + M2,  out(2t + 1) = M1 - M2 - M3
//
// so a (32-channel chunk) costs 12 (ky, xi) "taps" over HALF as many GEMM rows (a row is a 2-pixel tile) instead of
// 9 taps over all pixels: 1.5x fewer MFMAs.  (The 2-D form F(2x2, 3x3) saves 2.25x but needs 16 accumulators per
// output tile: a quarter of the GEMM rows per workgroup at the same register budget, and 16 transformed filter
// slices per chunk -- 57-77 B/clk/CU of operand DMA at full MFMA rate, three times what the L2 -> LDS path sustains
// in these kernels.  The 1-D form keeps conv_t32's weight bytes per MFMA.)  What is rounded that a direct convolution
// does not round: V (one f16 subtraction of two f16 values: exact to half an ulp) and U (once); tools/winograd_gate.py
// restates this arithmetic inside the f16-emulating oracle: every stage stays within 0.6x of the error budget.
//
// Structure (everything not said here is conv_t32's: weight slices pre-packed as LDS images, ring of R slices with
// counted vmcnt, fragments read one K-step ahead and across the barrier, hand-placed fillers, persistent tile walk):
//   * GEMM rows are tiles t = (n * H + y) * W / 2 + x / 2; a workgroup owns BMT consecutive tiles x BN channels and
//     keeps FOUR accumulator sets (one per xi) per wave tile: 64 tiles x 96 channels = 384 VGPRs, one wave per SIMD;
//   * per chunk the raw pixel range [2 lo - 1, 2 (lo + v_rows) + 1) is DMA'd into LDS (while the previous chunk
//     computes), then a transform pass turns it into four V planes of v_rows x 64 bytes: lane (row r, 16-byte
//     channel group g) reads pixels 2r .. 2r + 3, masks d0 / d3 at the left / right image border (that IS the zero
//     padding along x), and writes one 16-byte piece per plane;
//   * tap (ky, xi) reads plane xi at row shift (ky - 1) * W / 2, rows above / below the image through the zero
//     block (lane masks in SGPR pairs); no left / right masks are left in the K loop;
//   * the epilogue forms the two output pixels of every tile and runs the shared bias / SiLU / shortcut / 16-byte
//     store code once for the even and once for the odd pixels (row stride 2).
//
// OUTCOME (round 3, MI355X; DESIGN.md "Winograd, measured"): the gate passes -- numerically this kernel may replace the
// direct one (tests/test_gpu_conv.py::test_conv_w1d_every_tile, test_gpu_network.py::test_network_on_the_winograd_kernels)
// -- and it is SLOWER: 396 vs 296 us on M409600 N192 K1728, 541 vs 350 us on M1638400 N96 K864.  -DRMR_W1D_ABLATE says why:
// without its epilogue it takes 258 / 294 us, without epilogue, transform and waits 211 / 236 us, i.e. the K loop keeps the
// matrix pipe 34-40 % busy where conv_t32's keeps it 61 % busy.  Four accumulator sets are 384 registers per wave tile:
// ONE wave per SIMD, so nothing runs beside a wave while it issues its 3 LDS-DMA and 10 ds_read_b128 instructions per 12
// MFMAs (conv_t32 runs two waves per SIMD, or 24 MFMAs per wave and tap); the raw range is twice the bytes per MFMA; and
// the epilogue of a lone workgroup is fully exposed (45 % of the launch).  Fewer MFMAs do not help a loop that is bound
// by issue slots and the L2 -> LDS path.  The kernel stays in the tree behind RMR_WINOGRAD=1 (off by default: the
// autotuner would never pick it).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include "conv_t32_common.h"

namespace rmr {

namespace {

using namespace t32;

constexpr int NT = 12;   // taps per chunk: 3 filter rows x 4 transformed columns

// WM x WN waves, MREP x NREP fragments of 32 x 32 per wave and xi, A_SLOTS raw-range DMA instructions per tap (the
// first NT - (R - 3) taps of a chunk carry the next chunk's range: the last of them is waited for by the end of
// tap NT - 1), R weight slices in the ring.
// ABL (timing experiments only; results are wrong): 1 = no vmcnt waits in the taps, 2 = no input transform, 4 = no epilogue,
// 8 = no MFMAs
template <int WM, int WN, int MREP, int NREP, int A_SLOTS, int R, int ABL = 0>
__global__ __launch_bounds__(WM* WN * 64, 1) void conv_w1d_kernel(const ConvArgs a, const int v_rows, const int n_tiles, const int prio) {
    constexpr int NW = WM * WN;
    constexpr int BMT = WM * MREP * 32;         // tiles (2 pixels each) per workgroup
    constexpr int BN = WN * NREP * 32;
    constexpr int NB = BN / 16;                 // weight DMA instructions per tap
    constexpr int SLOTS = NB + A_SLOTS;         // DMA instructions per tap (workgroup)
    constexpr int D = (SLOTS + NW - 1) / NW;    // per wave
    constexpr int SLOT_BYTES = BN * 64;         // one (chunk, tap) weight slice
    constexpr int ATAPS = NT - (R - 3);         // taps that carry raw-range blocks
    constexpr unsigned OOB = 0xffff0000u;
    static_assert(R >= 4 && R <= 6, "ring depth");
    static_assert((R - 3) * D <= 63, "vmcnt is 6 bits");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const unsigned lds0 = sgpr((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const int raw_rows = (2 * v_rows + 2 + 15) & ~15;
    const int raw_bytes = raw_rows * 64;
    const int plane = v_rows * 64;
    const int v_base = raw_bytes;
    const int ring_base = raw_bytes + 4 * plane;
    const int zero_off = ring_base + R * SLOT_BYTES;   // 64 zero bytes, head of the scratch KiB
    const int bias_off = zero_off + 1024;              // Cout_pad floats

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
        m0 = (lid / nt_count) * BMT;
        n0 = (lid % nt_count) * BN;
    };
    int vb = blockIdx.x;
    if (vb >= n_tiles) return;
    int m0, n0;   // m0: first TILE of the workgroup's range
    tile_m0n0(vb, m0, n0);
    const int W2 = a.W >> 1;
    const int npix = a.M;

    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu), sgpr(a.in_bytes), sgpr(0x00020000u)};
    const u32x4 wt_rsrc = {sgpr((unsigned)(size_t)a.wt_w1d), sgpr((unsigned)((size_t)a.wt_w1d >> 32) & 0xffffu), sgpr(a.wt_w1d_bytes),
                           sgpr(0x00020000u)};

    // ---- DMA constants of this lane ----------------------------------------------------------
    const int lrow = lane >> 2;                                   // row inside a 16-row DMA block
    const int lch = (lane & 3) ^ ((lrow >> 2) & 3);               // logical 16-byte chunk it fetches
    const unsigned cs2 = (unsigned)a.in_cs * 2u;
    const unsigned in_cb = (unsigned)((a.in_co + lch * 8) * 2);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int na = raw_rows / 16;                                 // raw-range DMA blocks per chunk
    const int chunks = a.Cin / 32;
    const int total = chunks * NT;
    const unsigned wstep = (unsigned)(a.Cout_pad / 16) * 1024u;   // bytes of one (chunk, tap) slice of all channels
    const unsigned scratch = sgpr(lds0 + zero_off);

    if (tid < 4) *(u32x4*)(smem + zero_off + tid * 16) = u32x4{0, 0, 0, 0};
    for (int i = tid; i < a.Cout_pad; i += NW * 64) *(float*)(smem + bias_off + i * 4) = a.bias[i];

    // byte offset of this lane's piece of raw block ia of a range whose LDS row 0 is pixel lo_l, channel chunk cc
    const auto in_off = [&](int lo_l, int ia, int cc) {
        const int p = min(max(lo_l + ia * 16, 0), npix - 1);  // out-of-range pixels only ever reach masked taps / unstored rows
        return __umul24((unsigned)p, cs2) + in_cb + (unsigned)cc * 64u;
    };

    // DMA slot j of this wave is q = wave + NW * j: a weight block (q < NB) or a raw-range block, for the whole kernel
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
        s_aidx[j] = q - NB;
    }

    // ---- cold start: the whole raw range of chunk 0, weight slices 0 .. R-2 of the first tile -----------
    {
        const int pl0 = 2 * (m0 - W2) - 1 + lrow;
        for (int ia = wave; ia < na; ia += NW) dma16s(in_rsrc, sgpr(lds0 + ia * 1024), in_off(pl0, ia, 0), 0u);
#pragma unroll
        for (int s = 0; s < R - 1; ++s)
            for (int q = wave; q < NB; q += NW)
                dma16s(wt_rsrc, sgpr(lds0 + ring_base + s * SLOT_BYTES + q * 1024), s < total ? lane16 : OOB,
                       sgpr((unsigned)s * wstep + (unsigned)(n0 / 16 + q) * 1024u));
    }

    // ---- fragment constants ------------------------------------------------------------------------
    const int fr = lane & 31, kq = lane >> 5;
    const int a_row0 = wm * MREP * 32 + fr + W2;   // V row of the centre filter row of fragment 0 (V row 0 is tile m0 - W2)
    int zsel[MREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i) zsel[i] = zero_off - i * 2048;
    const int wlane = ring_base + (wn * NREP * 32 + fr) * 64 + ((kq ^ ((fr >> 2) & 3)) << 4);
    const auto lds16 = [&](int off) { return *(const half8*)(smem + off); };
    // The wave's 24 accumulators are 384 registers: more than either register file of a lane holds (an instruction field
    // names 256 ArchVGPRs or 256 AccVGPRs), and hipcc selects ONE form for every MFMA of a function -- with the builtin
    // it parks a third of the accumulators in the other file and shuttles them through v_accvgpr_read / _write around
    // each MFMA (measured in the ISA: 700 spilled registers, 32 copies per MFMA).  So the MFMAs are written out: the
    // first sixteen accumulators live in AccVGPRs ("+a"), the last eight in ArchVGPRs ("+v"), for the whole kernel.
    // No VALU instruction touches an accumulator inside the K loop, so none of the MFMA <-> VALU hazard wait states
    // the compiler would insert are needed there; the two places where VALU meets accumulators (zeroing, the output
    // transform) are a transform pass / a barrier away from the nearest MFMA, and the second has explicit s_nops.
    const auto mma = [](auto Q, half8 w, half8 x, floatx16& c) {
        if constexpr (ABL & 8)
            asm volatile("" : "+v"(c) : "v"(w), "v"(x));
        else if constexpr (decltype(Q)::value < 16)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(x));
        else
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
    };
    // address of fragment 0's chunk for K-step 0 of filter row ky in plane 0
    const auto a_addr = [&](int ky) {
        const int row = a_row0 + (ky - 1) * W2;
        return v_base + row * 64 + ((kq ^ ((row >> 2) & 3)) << 4);
    };
    // transform: this lane's (row, channel group) of pass 0; a pass covers 64 rows
    const int t_r0 = tid >> 2, t_g = tid & 3;
    const int t_rows_per_pass = NW * 16;
    const int t_step = t_rows_per_pass % W2;

    wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    constexpr int NM = MREP * NREP;   // MFMAs per K-step
    int slot = 0;                     // ring slot of the tap being computed
    // the weight stream: slice gw of the tile whose channel-tile offset is w_tile is the next one to fetch
    unsigned gw = R - 1;
    unsigned gwoff = (unsigned)(R - 1) * wstep;
    unsigned w_tile = (unsigned)(n0 / 16) * 1024u;
    unsigned w_live = 1u;
    half8 wa[NREP], wb[NREP];
    int wcur = wlane;
#pragma unroll
    for (int j = 0; j < NREP; ++j) wa[j] = lds16(wcur + j * 2048);

    for (;;) {
        if (prio == 1) __builtin_amdgcn_s_setprio(1);
        // ---- this tile and the next one -------------------------------------------------------------
        const int vbn = vb + G;
        const bool has_next = vbn < n_tiles;
        int m0n = 0, n0n = 0;
        if (has_next) tile_m0n0(vbn, m0n, n0n);
        const int pl = 2 * (m0 - W2) - 1 + lrow, pln = 2 * (m0n - W2) - 1 + lrow;
        const unsigned w_tile_next = (unsigned)(n0n / 16) * 1024u;
        // rows above / below the image of this lane's tile, per fragment
        bool up[MREP], dn[MREP];
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            const int m = m0 + (wm * MREP + i) * 32 + fr;
            const int y = (m / W2) % a.H;
            up[i] = y > 0;
            dn[i] = y < a.H - 1;
        }
        // column of V row t_r0 inside its image row (the tile range starts W2 tiles before m0)
        int t_x = (m0 + t_r0) % W2;
        floatx16 acc[4][MREP][NREP];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[x][i][j][r] = 0.f;

        half8 xa[MREP], xb[MREP];
        int selx[MREP];

        for (int cc = 0; cc < chunks; ++cc) {
            // ---- transform: raw pixels -> four V planes (the raw range landed before the last barrier; nobody
            // reads the planes any more: every wave is past the previous chunk's last tap) ----
            if constexpr (!(ABL & 2)) {
                int tx = t_x;
                for (int r = t_r0; r < v_rows; r += t_rows_per_pass) {
                    const int q = 2 * r;
                    half8 d0 = lds16((q + 0) * 64 + ((t_g ^ (((q + 0) >> 2) & 3)) << 4));
                    const half8 d1 = lds16((q + 1) * 64 + ((t_g ^ (((q + 1) >> 2) & 3)) << 4));
                    const half8 d2 = lds16((q + 2) * 64 + ((t_g ^ (((q + 2) >> 2) & 3)) << 4));
                    half8 d3 = lds16((q + 3) * 64 + ((t_g ^ (((q + 3) >> 2) & 3)) << 4));
                    const half8 z = {};
                    d0 = tx == 0 ? z : d0;            // x = -1: the zero padding
                    d3 = tx == W2 - 1 ? z : d3;       // x = W
                    const int dst = v_base + r * 64 + ((t_g ^ ((r >> 2) & 3)) << 4);
                    *(half8*)(smem + dst) = d0 - d2;
                    *(half8*)(smem + dst + plane) = d1 + d2;
                    *(half8*)(smem + dst + 2 * plane) = d2 - d1;
                    *(half8*)(smem + dst + 3 * plane) = d1 - d3;
                    tx += t_step;
                    tx = tx >= W2 ? tx - W2 : tx;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // fragments of (tap 0, K-step 0)
            {
                const int at = a_addr(0);
#pragma unroll
                for (int i = 0; i < MREP; ++i) {
                    selx[i] = up[i] ? at : zsel[i];
                    xa[i] = lds16(selx[i] + i * 2048);
                }
            }
            // the range fetched during this chunk: the next chunk of this tile, or chunk 0 of the next tile
            const bool in_tile = cc + 1 < chunks;
            const bool a_live = in_tile || has_next;
            const int a_pl = in_tile ? pl : pln;
            const int a_cc = in_tile ? cc + 1 : 0;
            const auto tap = [&](auto T) {
                constexpr int t = decltype(T)::value;
                constexpr int xi = t % 4;
                constexpr int tn = (t + 1) % NT;
                constexpr int kyn = tn / 4, xin = tn % 4;
                const int slot_w = slot == 0 ? R - 1 : slot - 1;
                const unsigned wv = w_live ? lane16 : OOB;
                int at_n = 0;
                __builtin_amdgcn_s_barrier();
                // fillers of K-step 0: the K-step 1 fragments of this tap (tiles, then weights), D DMA slots, the
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
                        constexpr int d = f - 2;
                        constexpr bool all_w = NW * (d + 1) <= NB, all_a = NW * d >= NB;
                        constexpr bool a_tap = t < ATAPS;
                        const unsigned w_lds = s_wdst[d] + slot_w * SLOT_BYTES, w_soff = s_wsrc[d] + w_tile + gwoff;
                        const int ia = t * A_SLOTS + s_aidx[d];
                        const bool alive = a_tap && a_live && ia < na && s_aidx[d] < A_SLOTS;
                        const unsigned a_lds = alive ? lds0 + ia * 1024 : scratch;
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
                            asm volatile("" : "+v"(av));
                            av = (a_tap && alive) ? av : OOB;
                            dma16s(s_rsrc[d], sgpr(isw ? w_lds : a_lds), isw ? wv : av, sgpr(isw ? w_soff : 0u));
                        }
                    } else {
                        at_n = a_addr(kyn) + xin * plane;
                        const int slot_n = slot + 1 == R ? 0 : slot + 1;
                        wcur = wlane + slot_n * SLOT_BYTES;
                        slot = slot_n;
                    }
                };
                static_for<0, NM>([&](auto Kc) {
                    constexpr int k = decltype(Kc)::value;
                    mma(tap_c<xi * NM + k>{}, wa[k % NREP], xa[k / NREP], acc[xi][k / NREP][k % NREP]);
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, D + 3>([&](auto Fc) {
                        if constexpr (decltype(Fc)::value * NM / (D + 3) == k) filler0(Fc);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                static_for<0, NM>([&](auto Kc) {
                    constexpr int k = decltype(Kc)::value;
                    mma(tap_c<xi * NM + k>{}, wb[k % NREP], xb[k / NREP], acc[xi][k / NREP][k % NREP]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (k == 0 && tn != 0) {
                        // K-step 0 fragments of the next tap: legal before the next barrier (the planes are complete).  The
                        // chunk's last tap has nothing to read ahead: the next chunk's planes do not exist yet.
#pragma unroll
                        for (int i = 0; i < MREP; ++i) {
                            const bool v = kyn == 0 ? up[i] : kyn == 2 ? dn[i] : true;
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
                // DMAs issued R - 3 taps ago (and earlier) have landed
                constexpr int pending = [] {
                    int n = 0;
                    for (int k = 0; k < R - 3; ++k) {
                        const int tt = (t - k + NT) % NT;
                        for (int j = 0; j < D; ++j) n += (NW * j >= NB && tt >= ATAPS) ? 0 : 1;
                    }
                    return n;
                }();
                if constexpr (!(ABL & 1)) wait_vm<pending>();
                // advance the weight stream; behind a tile's last slice comes the first one of the next tile
                const unsigned wrap = 0u - (unsigned)(gw + 1 == (unsigned)total);
                gw = (gw + 1) & ~wrap;
                gwoff = (gwoff + wstep) & ~wrap;
                w_tile ^= (w_tile ^ w_tile_next) & wrap;
                w_live ^= (w_live ^ (unsigned)has_next) & wrap;
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
            tap(tap_c<9>{});
            tap(tap_c<10>{});
            tap(tap_c<11>{});
            // every wave is past its last read of the planes, and this wave's raw blocks have landed
            __builtin_amdgcn_s_barrier();
        }

        // ---- output transform + epilogue; the next tile's first slices and raw range are in flight meanwhile
        if (prio == 1) __builtin_amdgcn_s_setprio(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results (hand-written MFMAs: no compiler hazard handling)
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j) {
                const floatx16 m1 = acc[1][i][j], m2 = acc[2][i][j];
                acc[0][i][j] = acc[0][i][j] + m1 + m2;
                acc[1][i][j] = m1 - m2 - acc[3][i][j];
            }
        if constexpr (ABL & 4) {
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" : : "v"(acc[0][i][j]), "v"(acc[1][i][j]));
#endif
                }
        } else {
            epilogue<MREP, NREP, 0, true, true, false, 0, 2>(a, acc[0], smem, 0, bias_off, m0, n0, wm, wn, lane, 0);
            epilogue<MREP, NREP, 0, true, true, false, 0, 2>(a, acc[1], smem, 0, bias_off, m0, n0, wm, wn, lane, 1);
        }

        if (!has_next) break;
        vb = vbn;
        m0 = m0n;
        n0 = n0n;
    }
    wait_vm<0>();
}

struct W1dTile {
    int bmt, bn, threads, a_slots, ring;
    void (*kernel)(const ConvArgs, int, int, int);
};

#define W1D(WM, WN, MR, NR, AS, R) \
    { WM * MR * 32, WN * NR * 32, WM * WN * 64, AS, R, conv_w1d_kernel<WM, WN, MR, NR, AS, R> }
#define W1DA(WM, WN, MR, NR, AS, R, ABL) \
    { WM * MR * 32, WN * NR * 32, WM * WN * 64, AS, R, conv_w1d_kernel<WM, WN, MR, NR, AS, R, ABL> }

// one four-wave workgroup per CU, one wave per SIMD: 4 xi x (64 tiles x 96 channels) = 384 accumulator registers per wave
const W1dTile kW1dTiles[] = {
    W1D(4, 1, 2, 3, 6, 4),   // 0: 256 tiles (512 pixels) x 96 channels: every map width up to 80
    W1D(2, 2, 2, 3, 4, 4),   // 1: 128 tiles (256 pixels) x 192 channels: the input range is read once for 192 channels
    W1D(4, 1, 2, 2, 4, 4),   // 2: 256 tiles x 64 channels (Detect box branch)
    // (rings of 5 and 6 slices -- two / three taps for a DMA to land -- measured the same as 4: 550 / 384 vs 541 / 396 us)
#ifdef RMR_W1D_ABLATE
    W1DA(4, 1, 2, 3, 6, 4, 1),    // 3: tile 0 without vmcnt waits
    W1DA(4, 1, 2, 3, 6, 4, 2),    // 4: without the transform
    W1DA(4, 1, 2, 3, 6, 4, 4),    // 5: without the epilogue
    W1DA(4, 1, 2, 3, 6, 4, 7),    // 6: MFMAs, DMA issue, fragment reads, barriers only
#endif
};
constexpr int kNumW1dTiles = sizeof(kW1dTiles) / sizeof(kW1dTiles[0]);

int w1d_v_rows(int bmt, int W) { return (bmt + W + 15) / 16 * 16; }   // BMT + 2 * (W / 2) tiles
int w1d_raw_rows(int v_rows) { return (2 * v_rows + 2 + 15) / 16 * 16; }
int w1d_lds_bytes(const W1dTile& t, int W, int cout_pad) {
    const int v = w1d_v_rows(t.bmt, W);
    return w1d_raw_rows(v) * 64 + 4 * v * 64 + t.ring * t.bn * 64 + 1024 + cout_pad * 4;
}

}  // namespace

int conv_w1d_num_tiles() { return kNumW1dTiles; }
ConvTile conv_w1d_tile(int id) { return ConvTile{kW1dTiles[id].bmt * 2, kW1dTiles[id].bn, 32}; }

bool conv_w1d_supported(const ConvArgs& a, int tile) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin % 32 || a.Cin < 32) return false;
    if (a.Ho != a.H || a.Wo != a.W || (a.W & 1) || a.W < 4 || a.pre || a.in_slab_c || a.out_slab_c || !a.wt_w1d) return false;
    if (tile < 0) return true;
    const W1dTile& t = kW1dTiles[tile];
    const int na = w1d_raw_rows(w1d_v_rows(t.bmt, a.W)) / 16;
    return a.Cout_pad % t.bn == 0 && na <= t.a_slots * (NT - (t.ring - 3)) && w1d_lds_bytes(t, a.W, a.Cout_pad) <= 160 * 1024;
}

void launch_conv_w1d(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile) {
    if (tile < 0 || tile >= kNumW1dTiles) fail(RMR_ERR_INVALID_ARGUMENT, "conv_w1d: tile %d out of range", tile);
    if (!conv_w1d_supported(a, tile)) fail(RMR_ERR_LOGIC, "conv_w1d: layer not supported by tile %d", tile);
    const W1dTile& t = kW1dTiles[tile];
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 4 || a.out_co % 4) fail(RMR_ERR_LOGIC, "conv_w1d: misaligned view");
    if (a.in_bytes == 0 || a.in_bytes > 0xf0000000ull || a.wt_w1d_bytes == 0)
        fail(RMR_ERR_LOGIC, "conv_w1d: buffer sizes not set or input view larger than 3.75 GiB");
    static std::once_flag once;
    std::call_once(once, [] {
        for (const W1dTile& d : kW1dTiles)
            (void)hipFuncSetAttribute((const void*)d.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int v_rows = w1d_v_rows(t.bmt, a.W);
    const int lds = w1d_lds_bytes(t, a.W, a.Cout_pad);
    const int mt = a.M / 2;
    const int n_tiles = ((mt + t.bmt - 1) / t.bmt) * (a.Cout_pad / t.bn);
    const int grid = std::min((n_tiles + 7) / 8 * 8, ctx.num_cus);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;   // the layer's own count, not the MFMA work
    const double bytes = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d q%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, tile);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    t.kernel<<<grid, t.threads, lds, stream>>>(a, v_rows, n_tiles, 0);
    RMR_HIP(hipGetLastError());
}

// [Cout_pad][Kp] (k = (ky * 3 + kx) * Cin + ci, f16) -> U = g G^T per filter row, rounded to f16 once, as the LDS images of
// the 12 (ky, xi) slices of every 32-channel chunk (pack_conv_weights_t32's layout with taps = 12)
void pack_conv_weights_w1d(const __half* packed, int cout_pad, int cin, int Kp, std::vector<__half>& out) {
    std::vector<__half> u((size_t)cout_pad * NT * cin);
    for (int n = 0; n < cout_pad; ++n)
        for (int ky = 0; ky < 3; ++ky)
            for (int c = 0; c < cin; ++c) {
                const float g0 = __half2float(packed[(size_t)n * Kp + (size_t)(ky * 3 + 0) * cin + c]);
                const float g1 = __half2float(packed[(size_t)n * Kp + (size_t)(ky * 3 + 1) * cin + c]);
                const float g2 = __half2float(packed[(size_t)n * Kp + (size_t)(ky * 3 + 2) * cin + c]);
                const float uu[4] = {g0, (g0 + g1 + g2) * 0.5f, (g0 - g1 + g2) * 0.5f, g2};
                for (int xi = 0; xi < 4; ++xi) u[(size_t)n * NT * cin + (size_t)(ky * 4 + xi) * cin + c] = __float2half(uu[xi]);
            }
    pack_conv_weights_t32(u.data(), cout_pad, cin, NT * cin, out, NT);
}

}  // namespace rmr
