// conv_sb.hip -- convolutions of SMALL batches (1-8 images: the batch-1 frame of BASELINE configs[1], one car image and a
// handful of armor crops) on v_mfma_f32_32x32x16_f16.  Replaces, for those launches, the layers of the TensorRT engine
// behind /root/reference/src/detect/detector.h:122.
//
// Why another family.  The throughput kernels (conv_t32 / conv_halo / conv_dma) are built for launches that fill the chip
// for hundreds of microseconds.  A layer of a batch-1 frame is 0.5 us of MFMA work for the chip and took them 10-17 us; stamps
// inside this kernel's first versions (profiles/r05_sb_stamps.txt, DESIGN.md section 4 "Batch 1") showed what those
// microseconds are: not launches, not L2 misses (operands hot or cold in the L2: the same time; every DMA out of range: the
// same timeline), but the INSTRUCTION STREAM of a workgroup's few waves -- a lone wave issues an instruction every 4-8 cycles,
// a taken branch costs ~10 ns, dependent MFMAs wait 16 passes for each other -- and the CU's address unit, which takes one
// LDS-DMA instruction per 16 cycles and holds the issuing wave meanwhile.  So:
//
//   * a STAGE is the unit of arrival: one 32-channel chunk with all nine taps and the tile's input range (halo form,
//     3x3 / stride 1: the range [m0 - W - 1, m0 + BM + W + 1) staged once, taps as row shifts of the fragment reads), or a
//     few (tap, chunk) units with their gathered pixel rows (gathered form: 1x1 layers and strided 3x3 layers, padding
//     as out-of-range DMA offsets that arrive as zeros).  The ring holds as many stages as LDS allows -- all of them for
//     most layers of a batch-1 frame;
//   * loader waves issue the DMAs (tables decided once per tile: no branches, no divisions in the issue code), one stage
//     ahead of the stage they wait for with a counted vmcnt chosen at run time (64-way switch), and report a landed stage
//     through the workgroup barrier; the other waves only compute;
//   * small tiles (32 x 32 ... 128 x 96 outputs) so that a batch-1 layer has 100-600 workgroups; the WK waves of a wave tile
//     share its K range (unit g of a stage goes to wave g % WK), a stage's fragment reads are issued ahead of its MFMAs, which
//     run on independent accumulator chains; the partial tiles meet in LDS, summed in wave order (deterministic), and two of
//     the waves share the epilogue;
//   * weights are read from the LDS images conv_t32 / conv_g32 already keep ([chunk][tap][Cout / 16][64 lanes][8],
//     pack_conv_weights_t32): a weight DMA is one contiguous KiB; same row swizzle, same fragment reads, same epilogue
//     arithmetic (conv_t32_common.h);
//   * several INDEPENDENT layers can leave in one launch (conv_sb_group_kernel: the Detect head's branches).
//
// The kernel body is a device function of (tile, layer): the stand-alone launch runs one tile per workgroup.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include "conv_t32_common.h"

namespace rmr {

namespace {

using namespace t32;

struct SbGeom {
    int taps;          // 9 or 1
    int a_rows;        // halo form: rows of the input range (multiple of 16)
    int units;         // (tap, chunk) units per stage (halo form: 9)
    int stage_bytes;
    int ns;            // ring slots
    int stages;        // stages of a tile
    int total_units;   // taps * chunks
    int mt, nt;        // tiles along M and along N
    int n_inner;       // consecutive tile ids differ in the channel tile (1) or in the pixel tile (0)
    int zero_off;      // LDS offset of one KiB of zeros
    int ablate;        // development builds (-DRMR_SB_TIMING): parts switched off, timing only
};

// LDS-DMA with the wait states a freshly written scalar operand needs in front of a vector-memory instruction
__device__ __forceinline__ void dma_sb(u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);   // wave-uniform by construction; the compiler cannot always tell
    soff = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
    asm volatile("s_nop 3\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

// s_waitcnt vmcnt(n) for a run-time n (clamped to the counter's 6 bits: waiting for fewer is always safe)
__device__ __forceinline__ void wait_vm_dyn(int n) {
#define SB_W(k) \
    case k:     \
        asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); \
        break;
    switch (n) {
        SB_W(0) SB_W(1) SB_W(2) SB_W(3) SB_W(4) SB_W(5) SB_W(6) SB_W(7) SB_W(8) SB_W(9) SB_W(10) SB_W(11) SB_W(12) SB_W(13) SB_W(14) SB_W(15)
        SB_W(16) SB_W(17) SB_W(18) SB_W(19) SB_W(20) SB_W(21) SB_W(22) SB_W(23) SB_W(24) SB_W(25) SB_W(26) SB_W(27) SB_W(28) SB_W(29) SB_W(30) SB_W(31)
        SB_W(32) SB_W(33) SB_W(34) SB_W(35) SB_W(36) SB_W(37) SB_W(38) SB_W(39) SB_W(40) SB_W(41) SB_W(42) SB_W(43) SB_W(44) SB_W(45) SB_W(46) SB_W(47)
        SB_W(48) SB_W(49) SB_W(50) SB_W(51) SB_W(52) SB_W(53) SB_W(54) SB_W(55) SB_W(56) SB_W(57) SB_W(58) SB_W(59) SB_W(60) SB_W(61) SB_W(62)
        default:
            asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
            break;
    }
#undef SB_W
}

// One output tile: BM = WM * MREP * 32 pixels x BN = WN * NREP * 32 channels; WK waves share each wave tile's K range.
// GATHER: false = halo form (3x3 / stride 1 / pad 1), true = gathered form (KH = KW = 1 or 3, any stride, pad = KH / 2).
// KM: units of a stage one wave takes (wave wk: units wk, wk + WK, ...: a stage has at most KM * WK units; halo form: 9).
//
// What the stamps of the first versions said (profiles/r05_sb_stamps.txt; M1600 N192 K1728, 64 x 32 tiles, 9.8 us per launch):
// the launch is paced by the INSTRUCTION STREAM of a workgroup's few waves, not by data -- with every DMA out of range
// and every MFMA removed the timeline does not move.  A wave issues an instruction every 4-8 cycles, so
//   * the K loop of a stage is ONE basic block: the fragments of up to GS units are read first, then their MFMAs run back to
//     back (one unit at a time on one accumulator was a chain of LDS latency + two dependent MFMAs per unit: 0.76 us per
//     stage of ten MFMAs).  A unit that does not exist (9 taps over WK = 4 waves; the last stage of a gathered layer) reads
//     the zero block instead of branching; where a wave tile is a single fragment the two K-steps of a unit accumulate into
//     separate registers (two independent MFMA chains);
//   * the DMA issue code has no branches and no divisions: what a wave's i-th DMA of a stage fetches is decided once per tile
//     (tables in scalar registers, per-lane gather offsets in vector registers), a stage adds its chunk offset; every wave
//     issues the same number of DMAs per stage (the surplus ones are out-of-range loads into the zero KiB), so the counted
//     wait behind a stage is a multiple of one wave-uniform number (a loop of scalar branches had cost 170 cycles per DMA);
//   * bias and shortcut are fetched BEFORE the first DMA (loads retire in order: they have landed with stage 0), the
//     epilogue is arithmetic and stores.
// LW: 0 = every wave issues its share of the DMAs and computes; 4 / 8 = the first LW waves of the workgroup (one / two per
// SIMD) only issue DMAs, one stage ahead of the stage they wait for, and the other WM * WN * WK waves only compute.  A wave gets
// an LDS-DMA instruction out every ~130 cycles and is held at the CU's address unit meanwhile (which takes one per 16 cycles):
// in the symmetric form every stage's MFMAs wait behind that stage's share of the issue (0.25 us); a loader wave held there
// leaves its SIMD's issue slots to the computing wave beside it (M1600 N192 K1728: 9.1 -> 7.4 us per launch).
template <int WM, int WN, int WK, int MREP, int NREP, bool GATHER, int KM, int LW>
__device__ __forceinline__ void sb_tile(const ConvArgs& a, const SbGeom& g, const int m0, const int n0, unsigned char* smem, const unsigned lds0) {
    constexpr int NC = WM * WN * WK;          // computing waves
    constexpr int NW = LW ? LW : NC;          // waves that issue DMAs
    constexpr int NT = LW + NC;               // waves of the workgroup
    constexpr int BM = WM * MREP * 32;
    constexpr int BN = WN * NREP * 32;
    constexpr int NB = BN / 16;   // weight DMA blocks per unit
    constexpr int PB = BM / 16;   // gathered form: pixel DMA blocks per unit
    constexpr unsigned OOB = 0xffff0000u;
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef RMR_SB_TIMING
    // development build: sixteen 100 MHz stamps per workgroup (wave 0; 8.. = when stage 0.. had landed), printed by rmr_conv_bench
    // under RMR_CONV_TIMING=1; RMR_SB_ABLATE (bit 0: no fragment reads / MFMAs, 1: weight DMAs out of range, 2: input DMAs out
    // of range) arrives in g.ablate
    const auto stamp = [&](int k) {
        // the DMA side's stamps (0, 1, 14, 15) by wave 0, the others by the first computing wave
        if (a.timing && tid == ((k == 0 || k == 1 || k >= 14) ? 0 : LW * 64)) a.timing[(size_t)blockIdx.x * 16 + k] = (long long)__builtin_amdgcn_s_memrealtime();
    };
    const unsigned in_lim = (g.ablate & 4) ? 0u : a.in_bytes, wt_lim = (g.ablate & 2) ? 0u : a.wt_t32_bytes;
#else
    const auto stamp = [](int) {};
    const unsigned in_lim = a.in_bytes, wt_lim = a.wt_t32_bytes;
#endif
    stamp(0);
    const bool loader = LW > 0 && wave < LW;   // a wave that only issues DMAs
    const int cw = loader ? 0 : wave - LW;      // the computing wave's index
    const int wk = cw % WK, wmn = cw / WK;
    const int wm = wmn / WN, wn = wmn % WN;
    const int W = a.W;
    const int npix = a.N * a.H * a.W;
    const int fr = lane & 31, kq = lane >> 5;

    // ---- the epilogue's loads first (the waves that run it: wk == 0) ----------------------------------------------------------
    // wide path: f16 output, SiLU, 8-channel aligned views (every layer of the backbone and neck); everything else (f32 logits,
    // no activation) leaves through the shared 8-byte epilogue, which fetches what it needs itself
    const bool wide = !a.out32 && a.act && ((a.out_cs | a.out_co) & 7) == 0 && (!a.res || ((a.res_cs | a.res_co) & 7) == 0);
    const int mw0 = m0 + wm * MREP * 32, nw0 = n0 + wn * NREP * 32;   // the wave tile's first pixel row and channel
    const long rows_left = (long)a.M - mw0;
    const auto view_bytes = [&](int cs) {
        const long b = rows_left * cs * 2;
        return (unsigned)(b <= 0 ? 0 : b > 0xfffffff0l ? 0xfffffff0l : b);
    };
    // with K shared by two or more waves the wide epilogue is shared as well: wave wk = 0 / 1 of a wave tile finishes channel
    // groups 0-1 / 2-3 (gp = 0 / 1) of every fragment
    constexpr int EW = WK >= 2 ? 2 : 1;        // waves of a wave tile that run the wide epilogue
    constexpr int GPN = EW == 2 ? 1 : 2;       // 16-channel halves of a fragment each of them finishes
    const int gp0 = EW == 2 ? wk : 0;
    float4 bias[NREP][GPN * 2];
    u32x4 rres[MREP][NREP][GPN];
    // (wave tiles of three and more fragments fetch them at the epilogue instead: 72 registers held through the K loop would
    // leave room for the fragments of one unit at a time)
    constexpr bool PRELOAD = MREP * NREP <= 2;
    const auto load_epilogue_operands = [&] {
#pragma unroll
        for (int j = 0; j < NREP; ++j)
#pragma unroll
            for (int gq = 0; gq < GPN * 2; ++gq) bias[j][gq] = *(const float4*)(a.bias + nw0 + j * 32 + (gp0 * 2 + gq) * 8 + kq * 4);
        if (a.res) {
            const __amdgpu_buffer_rsrc_t res_rsrc =
                __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)a.res + (long)mw0 * a.res_cs), 0, view_bytes(a.res_cs), 0x00020000);
            const unsigned res_lane = (unsigned)(fr * a.res_cs + a.res_co + nw0 + kq * 8) * 2u;
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j)
#pragma unroll
                    for (int gg = 0; gg < GPN; ++gg)
                        rres[i][j][gg] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, res_lane + (unsigned)(i * 32 * a.res_cs + j * 32 + (gp0 + gg) * 16) * 2u, 0, 0);
        }
    };
    if (PRELOAD && wide && wk < EW && !loader) load_epilogue_operands();

    stamp(14);
    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu), sgpr(in_lim), sgpr(0x00020000u)};
    const u32x4 wt_rsrc = {sgpr((unsigned)(size_t)a.wt_t32), sgpr((unsigned)((size_t)a.wt_t32 >> 32) & 0xffffu), sgpr(wt_lim), sgpr(0x00020000u)};

    // ---- DMA roles of this wave: row lane >> 2 of a 16-row block, 16-byte chunk (lane & 3) ^ key(row) ---------------------------
    const int lrow = lane >> 2;
    const int lch = (lane & 3) ^ ((lrow >> 2) & 3);
    const unsigned cs2 = (unsigned)a.in_cs * 2u;
    const unsigned in_cb = (unsigned)((a.in_co + lch * 8) * 2);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int nblk_all = a.Cout_pad / 16;
    const int n0_16 = n0 / 16;
    const int a_bytes = GATHER ? 0 : g.a_rows * 64;
    const unsigned zero_lds = lds0 + (unsigned)g.zero_off;   // one KiB of zeros: what masked taps read, where surplus DMAs land
    constexpr int UNIT_BYTES = (BM + BN) * 64;              // gathered form: pixel rows, then weight rows
    const int units_last = g.total_units - (g.stages - 1) * g.units;

    // halo form: WCNT weight blocks per wave and stage (block wq = wave + i * NW of the stage's 9 * NB: tap wq / NB, 16-channel
    // block wq % NB), then na_w input-range blocks (ia = wave + i * NW)
    constexpr int WCNT = GATHER ? 1 : (9 * NB + NW - 1) / NW;
    unsigned w_src[WCNT], w_dst[WCNT];
    unsigned w_voff_last = OOB;
    const int na = GATHER ? 0 : g.a_rows / 16;
    const int na_w = (na + NW - 1) / NW;        // wave-uniform: the surplus ones are out of range
    const int pbase = m0 - W - 1 + lrow;        // the pixel LDS row `lrow` of input block 0 holds
    // gathered form: GCNT blocks per wave and stage (block q = wave + i * NW: unit q / (PB + NB), row block q % (PB + NB) of it:
    // pixel rows first, then weight rows)
    constexpr int GCNT = GATHER ? (KM * WK * (PB + NB) + NW - 1) / NW : 1;
    int g_unit[GCNT];
    unsigned g_dst[GCNT], g_wsrc[GCNT];
    bool g_pix[GCNT];
    unsigned g_base[GCNT];   // per lane: byte offset of the pixel's tap (0, 0), channel chunk 0
    unsigned g_ok[GCNT];     // per lane: bit 3 ky + kx = that tap lies inside the image
    // (the waves that never issue a DMA skip this: LW > 0 and not a loader)
#pragma unroll
    for (int i = 0; i < WCNT; ++i) w_src[i] = 0u, w_dst[i] = 0u;
#pragma unroll
    for (int i = 0; i < GCNT; ++i) g_unit[i] = 0, g_dst[i] = 0u, g_wsrc[i] = 0u, g_pix[i] = false, g_base[i] = 0u, g_ok[i] = 0u;
    if (LW == 0 || loader) {
        if constexpr (!GATHER) {
#pragma unroll
            for (int i = 0; i < WCNT; ++i) {
                const int wq = wave + i * NW;
                const int t = wq / NB, b = wq - t * NB;
                w_src[i] = (unsigned)(t * nblk_all + n0_16 + b) * 1024u;
                w_dst[i] = wq < 9 * NB ? (unsigned)(a_bytes + wq * 1024) : 0u;
                if (i == WCNT - 1) w_voff_last = wq < 9 * NB ? lane16 : OOB;
            }
        } else {
#pragma unroll
            for (int i = 0; i < GCNT; ++i) {
                const int q = wave + i * NW;
                const int u = q / (PB + NB), r = q - u * (PB + NB);
                g_unit[i] = u;
                g_pix[i] = r < PB;
                g_dst[i] = (unsigned)(u * UNIT_BYTES + r * 1024);
                g_wsrc[i] = (unsigned)(n0_16 + r - PB) * 1024u;
                const int m = m0 + r * 16 + lrow;
                int ox = m, oy = 0, img = 0;
                if (!(a.KH == 1 && a.stride == 1)) {
                    ox = m % a.Wo;
                    const int qq = m / a.Wo;
                    oy = qq % a.Ho, img = qq / a.Ho;
                }
                const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
                unsigned ok = 0;
                if (r < PB && m < a.M) {
                    if (a.KH == 1) {
                        ok = 1u;
                    } else {
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx)
                                if (iy0 + ky >= 0 && iy0 + ky < a.H && ix0 + kx >= 0 && ix0 + kx < W) ok |= 1u << (ky * 3 + kx);
                    }
                }
                g_ok[i] = ok;
                g_base[i] = (a.KH == 1 && a.stride == 1) ? (unsigned)m * cs2 + in_cb : (unsigned)((img * a.H + iy0) * W + ix0) * cs2 + in_cb;
            }
        }
    }
    const int d_stage = GATHER ? GCNT : WCNT + na_w;   // DMAs per wave and stage
    const unsigned slab_inv = a.in_slab_c ? 65536u / (unsigned)a.in_slab_c + 1u : 0u;   // c / in_slab_c = (c * slab_inv) >> 16 for c < 2048

    const auto issue = [&](int s, int slot) {
        const unsigned base = lds0 + (unsigned)(slot * g.stage_bytes);
        if constexpr (!GATHER) {
            const unsigned wchunk = (unsigned)(s * 9 * nblk_all) * 1024u;
#pragma unroll
            for (int i = 0; i < WCNT; ++i)
                dma_sb(wt_rsrc, i == WCNT - 1 && w_dst[i] == 0u ? zero_lds : base + w_dst[i], i == WCNT - 1 ? w_voff_last : lane16, wchunk + w_src[i]);
            const unsigned cchunk = in_cb + (unsigned)s * 64u;
            int ia = wave;
            for (int i = 0; i < na_w; ++i, ia += NW) {
                const int p = min(max(pbase + ia * 16, 0), npix - 1);   // pixels outside the tensor are only read by masked taps
                const bool live = ia < na;
                dma_sb(in_rsrc, live ? base + (unsigned)ia * 1024u : zero_lds, live ? (unsigned)p * cs2 + cchunk : OOB, 0u);
            }
        } else {
            const int nu = s == g.stages - 1 ? units_last : g.units;
            const int g0 = s * g.units;
#pragma unroll
            for (int i = 0; i < GCNT; ++i) {
                const int gu = g0 + g_unit[i];
                const bool live = g_unit[i] < nu;
                int cc = gu, t = 0;
                if (g.taps == 9) {
                    cc = (int)(((unsigned)gu * 7282u) >> 16);   // gu / 9 for gu < 16384
                    t = gu - cc * 9;
                }
                const int ky = (t * 11) >> 5, kx = t - ky * 3;
                // the chunk's 16-byte piece of this lane: channels c .. c + 7 of the input; with planar channel groups (slabs: a C2f's
                // chunks, 1x1 layers only) they live in slab c / in_slab_c -- slab_inv = 0 without slabs, so the same arithmetic
                // gives cc * 64 bytes into the pixel
                const unsigned c8 = (unsigned)cc * 32u + (unsigned)lch * 8u;
                const unsigned slab = (c8 * slab_inv) >> 16;
                const unsigned tap_off = (unsigned)(ky * W + kx) * cs2 + slab * a.in_slab_stride + (c8 - slab * (unsigned)a.in_slab_c) * 2u - (unsigned)lch * 16u;
                const unsigned voff = g_pix[i] ? (live && ((g_ok[i] >> t) & 1u) ? g_base[i] + tap_off : OOB) : (live ? lane16 : OOB);
                const unsigned soff = g_pix[i] ? 0u : (unsigned)(gu * nblk_all) * 1024u + g_wsrc[i];
                if (g_pix[i])
                    dma_sb(in_rsrc, live ? base + g_dst[i] : zero_lds, voff, 0u);
                else
                    dma_sb(wt_rsrc, live ? base + g_dst[i] : zero_lds, voff, soff);
            }
        }
    };

    stamp(15);
    // ---- the first two stages go in flight before anything else; the rest is topped up two stages ahead of the one being
    // waited for (the DMA instructions of a whole operand set take 1.2 us to ISSUE -- 16 cycles each at the CU's address unit --
    // so a wave that issued everything first started its first MFMA 2.8 us into the kernel)
    constexpr int LOOKAHEAD = 2;
    int issued = 0, islot = 0;   // stages issued so far, ring slot of the next one
    const auto top_up = [&](int limit) {
        while (issued < g.stages && issued <= limit) {
            issue(issued, islot);
            ++issued;
            islot = islot + 1 == g.ns ? 0 : islot + 1;
        }
    };
    if (LW == 0 || loader) top_up(min(LOOKAHEAD, g.ns) - 1);
    stamp(1);
    // the zero KiB (surplus DMAs write zeros there as well, which keeps it zero)
    for (int i = tid; i < 64; i += NT * 64) *(u32x4*)(smem + g.zero_off + i * 16) = u32x4{0, 0, 0, 0};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // written before this wave reaches the first barrier

    if constexpr (LW > 0) {
        if (loader) {
            // the loader's whole life: keep LOOKAHEAD stages ahead, report each stage to the computing waves through the barrier
            __builtin_amdgcn_s_barrier();   // (the zero KiB)
            // (stage s + 1 is in flight while stage s is waited for, so the address unit never idles; more stages ahead only
            // delay the report of stage 0: with three the first MFMA started 2.4 us into the kernel)
            for (int s = 0; s < g.stages; ++s) {
                wait_vm_dyn(min((issued - 1 - s) * d_stage, 63));
                __builtin_amdgcn_s_barrier();   // stage s has landed; the computing waves are done with stage s - 1
                top_up(min(s + LOOKAHEAD, s + g.ns - 1));
            }
            return;
        }
    }
    // ---- fragment constants ----------------------------------------------------------------------------------------------
    const int fkey = (kq ^ ((fr >> 2) & 3)) << 4;   // 16-byte slot of K-step 0 in a row whose index is fr modulo 16
    int arow[MREP];                                  // halo form: LDS row of the centre tap; gathered form: the pixel row
    unsigned amask[MREP];                            // halo form: valid taps of this lane's pixel (bit 3 ky + kx)
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int r = (wm * MREP + i) * 32 + fr;
        if constexpr (GATHER) {
            arow[i] = r * 64 + fkey;
            amask[i] = 0x1ffu;
        } else {
            arow[i] = r + W + 1;
            const int m = m0 + r;
            const int x = m % W, y = (m / W) % a.H;
            const unsigned rows = (y > 0 ? 0x007u : 0u) | 0x038u | (y < a.H - 1 ? 0x1c0u : 0u);
            const unsigned cols = (x > 0 ? 0x049u : 0u) | 0x092u | (x < W - 1 ? 0x124u : 0u);
            amask[i] = rows & cols;
        }
    }
    const int wrow = (wn * NREP * 32 + fr) * 64 + fkey;   // + j * 2048: fragment j of the weight rows

    // accumulator sets: four independent MFMA chains per wave where the registers allow (a 32x32x16 MFMA has a latency of 16
    // passes; chained on one accumulator the six MFMAs of a stage took 0.3 us).  Unit parity and K-step parity choose the set.
    // GS: units whose fragments are read together, ahead of their MFMAs -- as many as the wave's register budget holds.
    constexpr int MAXV = NT <= 8 ? 256 : NT <= 12 ? 168 : 128;   // (ArchVGPRs: the accumulators of these kernels never sit in AccVGPRs)
    constexpr int FRAGS = MREP * NREP;
    constexpr int NACC_WANT = FRAGS == 1 ? 4 : FRAGS < 4 ? 2 : 1;
    constexpr int OTHER = 40 + (FRAGS <= 2 ? NREP * 16 + FRAGS * 8 : 0) + (GATHER ? 2 * GCNT + 16 : (FRAGS > 2 ? 24 : 0));   // addresses and tables, the epilogue's bias and shortcut registers
    constexpr int NACC = NACC_WANT * FRAGS * 16 + OTHER + 2 * (MREP + NREP) * 8 <= MAXV ? NACC_WANT : NACC_WANT > 2 ? 2 : 1;
    // one register set holds GS units; with several groups per stage a second set lets the next group's reads travel under this
    // group's MFMAs (PIPE), where the budget has room for two sets of at least one unit... and the stage more than one group
    constexpr int UNIT_REGS = (MREP + NREP) * 8;
    constexpr int ROOM = MAXV - NACC * FRAGS * 16 - OTHER;
    constexpr int GS_CAP = 16 / (MREP + NREP) < 1 ? 1 : 16 / (MREP + NREP);
    constexpr int GS_ONE = ROOM / UNIT_REGS < 1 ? 1 : (ROOM / UNIT_REGS < GS_CAP ? ROOM / UNIT_REGS : GS_CAP);   // single set
    constexpr int GS_TWO = ROOM / (2 * UNIT_REGS) < GS_CAP ? ROOM / (2 * UNIT_REGS) : GS_CAP;                       // two sets
    constexpr bool PIPE = GS_ONE < KM && GS_TWO >= 1;
    constexpr int GS_RAW = PIPE ? GS_TWO : GS_ONE;
    constexpr int GS = GS_RAW < KM ? GS_RAW : KM;
    floatx16 acc[NACC][MREP][NREP];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

    stamp(13);
    if constexpr (LW > 0) __builtin_amdgcn_s_barrier();   // pairs with the loaders' first barrier
    // ---- the stages ----------------------------------------------------------------------------------------------------------
    int slot = 0;          // ring slot of stage s
    for (int s = 0; s < g.stages; ++s) {
        // a stage may be issued once every wave is done with the stage whose slot it takes: up to s + ns - 2 before the barrier
        // of iteration s, up to s + ns - 1 behind it
        if constexpr (LW == 0) {
            top_up(min(s + LOOKAHEAD, s + g.ns - 2));
            wait_vm_dyn(min((issued - 1 - s) * d_stage, 63));   // the stages behind s may still be in flight (d_stage DMAs of this wave each)
        }
        if (s == 0) stamp(12);
        __builtin_amdgcn_s_barrier();   // stage s has landed for every wave, and every wave is done with stage s - 1
        if (s == 0) stamp(2);
        if (s == g.stages - 1) stamp(3);
        if (s < 4) stamp(8 + s);
        if constexpr (LW == 0) top_up(min(s + LOOKAHEAD, s + g.ns - 1));
        const int sb = slot * g.stage_bytes;
        const int nu = GATHER ? (s == g.stages - 1 ? units_last : g.units) : 9;
#ifdef RMR_SB_TIMING
        if (g.ablate & 1) {
            slot = slot + 1 == g.ns ? 0 : slot + 1;
            continue;
        }
#endif
        // GS units at a time: every fragment read of a group is issued before its first MFMA (left alone the compiler keeps
        // one read in flight and every MFMA waits for the LDS latency of its own operands), and where a stage has several
        // groups the reads of group g + 1 are issued BEFORE the MFMAs of group g (two register sets: the MFMAs wait with a
        // counted lgkmcnt for their own group only)
        constexpr int NG = (KM + GS - 1) / GS;
        half8 x0[2][GS][MREP], x1[2][GS][MREP], w0[2][GS][NREP], w1[2][GS][NREP];
        const auto read_group = [&](auto GC) {
            constexpr int gi = decltype(GC)::value;
            constexpr int bsel = gi & 1;
#pragma unroll
            for (int kk = 0; kk < GS; ++kk) {
                const int k = gi * GS + kk;
                if (k >= KM) continue;
                const int uu = wk + k * WK;          // wave-uniform
                const bool live = uu < nu;
                const int u = live ? uu : 0;
                int wb;
                if constexpr (GATHER) {
                    const int ub = sb + u * UNIT_BYTES;
                    wb = ub + BM * 64 + wrow;
#pragma unroll
                    for (int i = 0; i < MREP; ++i) {
                        const int sel = live ? ub + arow[i] : g.zero_off;
                        x0[bsel][kk][i] = *(const half8*)(smem + sel);
                        x1[bsel][kk][i] = *(const half8*)(smem + (sel ^ 32));
                    }
                } else {
                    const int ky = (u * 11) >> 5, kx = u - ky * 3;
                    const int shift = (ky - 1) * W + (kx - 1);
                    wb = sb + a_bytes + u * (BN * 64) + wrow;
#pragma unroll
                    for (int i = 0; i < MREP; ++i) {
                        const int row = arow[i] + shift;
                        const int at = sb + row * 64 + ((kq ^ ((row >> 2) & 3)) << 4);
                        const int sel = (live && ((amask[i] >> u) & 1u)) ? at : g.zero_off;
                        x0[bsel][kk][i] = *(const half8*)(smem + sel);
                        x1[bsel][kk][i] = *(const half8*)(smem + (sel ^ 32));
                    }
                }
#pragma unroll
                for (int j = 0; j < NREP; ++j) {
                    w0[bsel][kk][j] = *(const half8*)(smem + wb + j * 2048);
                    w1[bsel][kk][j] = *(const half8*)(smem + ((wb + j * 2048) ^ 32));
                }
            }
        };
        read_group(tap_c<0>{});
        static_for<0, NG>([&](auto GC) {
            constexpr int gi = decltype(GC)::value;
            constexpr int bsel = gi & 1;
            if constexpr (PIPE && gi + 1 < NG) read_group(tap_c<gi + 1>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < GS; ++kk) {
                if (gi * GS + kk >= KM) continue;
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < NREP; ++j) {
                        constexpr int c0 = 0;
                        const int c = NACC == 4 ? 2 * ((gi * GS + kk) & 1) : c0;
                        acc[c][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[bsel][kk][j], x0[bsel][kk][i], acc[c][i][j], 0, 0, 0);
                    }
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < NREP; ++j) {
                        const int c = NACC == 4 ? 2 * ((gi * GS + kk) & 1) + 1 : NACC - 1;
                        acc[c][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[bsel][kk][j], x1[bsel][kk][i], acc[c][i][j], 0, 0, 0);
                    }
            }
            if constexpr (gi + 1 < NG) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!PIPE) read_group(tap_c<gi + 1>{});
            }
        });
        slot = slot + 1 == g.ns ? 0 : slot + 1;
    }
#pragma unroll
    for (int c = 1; c < NACC; ++c)
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][i][j][r] += acc[c][i][j][r];

    stamp(4);
    // ---- the K shares of a wave tile meet in LDS (the ring is free now): every wave leaves its partial tile there, the waves
    // that run the epilogue sum the shares of THEIR channel groups in wave order (deterministic)
    if constexpr (WK > 1) {
        constexpr int FRAG = MREP * NREP * 4096;   // bytes of one wave's accumulators
        __builtin_amdgcn_s_barrier();
        {
            unsigned char* const dst = smem + (wk * WM * WN + wmn) * FRAG + lane * 16;
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *(float4*)(dst + ((i * NREP + j) * 4 + r) * 1024) =
                            make_float4(acc[0][i][j][4 * r], acc[0][i][j][4 * r + 1], acc[0][i][j][4 * r + 2], acc[0][i][j][4 * r + 3]);
        }
        __builtin_amdgcn_s_barrier();
        if (wk >= (wide ? EW : 1)) return;
        // float4 r of a fragment = registers 4 r .. 4 r + 3 = channel group r: the wide epilogue's wave takes r = 2 gp0, 2 gp0 + 1
        const int r_lo = wide ? 2 * gp0 : 0, r_n = wide ? 2 * GPN : 4;
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    if (rr >= r_n) continue;
                    const int r = r_lo + rr;
                    float4 sum = *(const float4*)(smem + (0 * WM * WN + wmn) * FRAG + lane * 16 + ((i * NREP + j) * 4 + r) * 1024);
#pragma unroll
                    for (int z = 1; z < WK; ++z) {
                        const float4 p = *(const float4*)(smem + (z * WM * WN + wmn) * FRAG + lane * 16 + ((i * NREP + j) * 4 + r) * 1024);
                        sum.x += p.x, sum.y += p.y, sum.z += p.z, sum.w += p.w;
                    }
                    const int q = rr;   // (r is a run-time value for the wide epilogue's second wave: its sums go to the registers of half 0)
                    acc[0][i][j][4 * q] = sum.x, acc[0][i][j][4 * q + 1] = sum.y, acc[0][i][j][4 * q + 2] = sum.z, acc[0][i][j][4 * q + 3] = sum.w;
                }
    }
    stamp(5);
    if (!wide) {
        epilogue<MREP, NREP, 0, false, true>(a, acc[0], smem, 0, 0, m0, n0, wm, wn, lane);
    } else {
        if constexpr (!PRELOAD) load_epilogue_operands();
        // conv_t32_common.h's wide epilogue with its loads hoisted to the top of the kernel: bias, SiLU, (+ shortcut in f32),
        // one rounding, lane pairs (l, l + 32) exchange halves so that each stores 16 bytes (8 consecutive channels).  With K
        // shared (WK > 1) this wave's half sits in registers 0 .. 7 of every fragment, otherwise half gg in registers 8 gg ..
        const __amdgpu_buffer_rsrc_t out_rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)((_Float16*)a.out + (long)mw0 * a.out_cs), 0, view_bytes(a.out_cs), 0x00020000);
        const unsigned out_lane = (unsigned)(fr * a.out_cs + a.out_co + nw0 + kq * 8) * 2u;
        const bool has_res = a.res != nullptr;
        const unsigned oslab_inv = a.out_slab_c ? 65536u / (unsigned)a.out_slab_c + 1u : 0u;
        const __amdgpu_buffer_rsrc_t oslab_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0xfffffff0u, 0x00020000);   // (the launch checked the span)
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
#pragma unroll
                for (int gg = 0; gg < GPN; ++gg) {
                    float v[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 b = bias[j][gg * 2 + h];
                        const int r0 = (gg * 2 + h) * 4;
                        v[h * 4 + 0] = silu_t(acc[0][i][j][r0 + 0] + b.x);
                        v[h * 4 + 1] = silu_t(acc[0][i][j][r0 + 1] + b.y);
                        v[h * 4 + 2] = silu_t(acc[0][i][j][r0 + 2] + b.z);
                        v[h * 4 + 3] = silu_t(acc[0][i][j][r0 + 3] + b.w);
                    }
                    union {
                        u32x4 u;
                        _Float16 h[8];
                        unsigned w[4];
                    } o;
                    if (has_res) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[r]), __float_as_uint(v[4 + r]), false, false);
                            v[r] = __uint_as_float(sw[0]);
                            v[4 + r] = __uint_as_float(sw[1]);
                        }
                        union {
                            u32x4 u;
                            _Float16 h[8];
                        } rr;
                        rr.u = rres[i][j][gg];
#pragma unroll
                        for (int r = 0; r < 8; ++r) o.h[r] = (_Float16)(v[r] + (float)rr.h[r]);
                    } else {
                        union {
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
                    if (a.out_slab_c) {
                        // planar channel groups: the 8 channels of this store lie in slab ch / out_slab_c; rows past M must not
                        // land in the next slab, so they go out of range by hand
                        const unsigned ch = (unsigned)(nw0 + j * 32 + (gp0 + gg) * 16 + kq * 8);
                        const unsigned so = (ch * oslab_inv) >> 16;
                        const long row = (long)mw0 + i * 32 + fr;
                        const unsigned off = row < a.M ? so * a.out_slab_stride + (unsigned)(row * a.out_cs + a.out_co + (ch - so * (unsigned)a.out_slab_c)) * 2u : OOB;
                        __builtin_amdgcn_raw_buffer_store_b128(o.u, oslab_rsrc, off, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(o.u, out_rsrc, out_lane + (unsigned)(i * 32 * a.out_cs + j * 32 + (gp0 + gg) * 16) * 2u, 0, 0);
                    }
                }
    }
    stamp(6);
#ifdef RMR_SB_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(7);
#endif
}

template <int WM, int WN, int WK, int MREP, int NREP, bool GATHER, int KM, int LW>
__global__ __launch_bounds__((WM * WN * WK + LW) * 64) void conv_sb_kernel(const ConvArgs a, const SbGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    // tile of this workgroup: the workgroups of one XCD (blockIdx & 7) take a contiguous range of tile ids, so that the tiles
    // that share input rows (n_inner) or weight rows share an L2
    const int n_tiles = g.mt * g.nt;
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
    const int xcd = blockIdx.x & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int mi = g.n_inner ? lid / g.nt : lid % g.mt, ni = g.n_inner ? lid % g.nt : lid / g.mt;
    sb_tile<WM, WN, WK, MREP, NREP, GATHER, KM, LW>(a, g, mi * (WM * MREP * 32), ni * (WN * NREP * 32), smem, lds0);
}

// Several INDEPENDENT layers in one launch (the Detect head: the same depth of its six branches): problem p owns the tile ids
// [tile0, tile0 + mt * nt).  A batch-1 layer is a fixed cost of launch, ramp and epilogue around very little arithmetic, so six
// of them side by side cost little more than the largest one.
struct SbProblem {
    ConvArgs a;
    SbGeom g;
    int tile0;
    int pad[3];
};

template <int WM, int WN, int WK, int MREP, int NREP, bool GATHER, int KM, int LW>
__global__ __launch_bounds__((WM * WN * WK + LW) * 64) void conv_sb_group_kernel(const SbProblem* __restrict__ probs, const int n_probs, const int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
    const int xcd = blockIdx.x & 7;
    const int gid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    int p = 0;
    for (int i = 1; i < n_probs; ++i)
        if (gid >= probs[i].tile0) p = i;
    const ConvArgs a = probs[p].a;
    const SbGeom g = probs[p].g;
    const int lid = gid - probs[p].tile0;
    const int mi = g.n_inner ? lid / g.nt : lid % g.mt, ni = g.n_inner ? lid % g.nt : lid / g.mt;
    sb_tile<WM, WN, WK, MREP, NREP, GATHER, KM, LW>(a, g, mi * (WM * MREP * 32), ni * (WN * NREP * 32), smem, lds0);
}

struct SbVariant {
    int bm, bn, wk, km, threads, wgs_per_cu, loaders;
    bool gather;
    void (*kernel)(const ConvArgs, const SbGeom);
    void (*group_kernel)(const SbProblem*, int, int);
};

// halo form: KM = ceil(9 / WK) taps per wave and stage; gathered form: KG units per wave and stage (a stage = KG * WK units)
#define SB3(WM, WN, WK, MR, NR, KG, WPC, LW)                                                                                                          \
    {WM * MR * 32, WN * NR * 32, WK, (9 + WK - 1) / WK, (WM * WN * WK + LW) * 64, WPC, LW, false, conv_sb_kernel<WM, WN, WK, MR, NR, false, (9 + WK - 1) / WK, LW>, conv_sb_group_kernel<WM, WN, WK, MR, NR, false, (9 + WK - 1) / WK, LW>}, \
    {WM * MR * 32, WN * NR * 32, WK, KG, (WM * WN * WK + LW) * 64, WPC, LW, true, conv_sb_kernel<WM, WN, WK, MR, NR, true, KG, LW>, conv_sb_group_kernel<WM, WN, WK, MR, NR, true, KG, LW>}
#define SB2(WM, WN, WK, MR, NR, KG, WPC) SB3(WM, WN, WK, MR, NR, KG, WPC, 0)

// even ids: halo form, odd ids: the gathered form of the same tile
const SbVariant kSbVariants[] = {
    SB2(1, 1, 3, 1, 1, 3, 1),   //  0 /  1:  32 x 32, three waves (three taps each)
    SB2(2, 1, 3, 1, 1, 3, 1),   //  2 /  3:  64 x 32, six waves
    SB2(1, 2, 3, 1, 1, 3, 1),   //  4 /  5:  32 x 64, six waves
    SB2(2, 2, 3, 1, 1, 2, 1),   //  6 /  7:  64 x 64, twelve waves
    SB2(4, 1, 3, 1, 1, 2, 1),   //  8 /  9: 128 x 32, twelve waves
    SB2(2, 1, 4, 1, 1, 2, 1),   // 10 / 11:  64 x 32, eight waves (nine taps over four waves: a quarter of the MFMAs are idle)
    SB2(2, 1, 2, 1, 1, 4, 1),   // 12 / 13:  64 x 32, four waves
    SB2(2, 1, 3, 2, 1, 2, 1),   // 14 / 15: 128 x 32, six waves, 64 x 32 per wave
    SB2(2, 2, 3, 2, 1, 2, 1),   // 16 / 17: 128 x 64, twelve waves
    SB2(2, 2, 2, 2, 1, 2, 1),   // 18 / 19: 128 x 64, eight waves
    SB2(2, 2, 1, 2, 1, 4, 1),   // 20 / 21: 128 x 64, four waves
    SB2(4, 1, 1, 1, 3, 4, 1),   // 22 / 23: 128 x 96, four waves, 32 x 96 per wave
    SB2(4, 1, 2, 1, 3, 2, 1),   // 24 / 25: 128 x 96, eight waves
    SB2(2, 1, 6, 1, 1, 2, 1),   // 26 / 27:  64 x 32, twelve waves (two taps per wave and stage, three of twelve idle)
    SB2(4, 2, 2, 1, 1, 2, 1),   // 28 / 29: 128 x 64, sixteen waves
    // the same tiles with half the LDS: two workgroups per CU
    SB2(1, 1, 3, 1, 1, 3, 2),   // 30 / 31:  32 x 32
    SB2(2, 1, 3, 1, 1, 3, 2),   // 32 / 33:  64 x 32
    SB2(2, 2, 3, 1, 1, 2, 2),   // 34 / 35:  64 x 64
    SB2(2, 1, 3, 2, 1, 2, 2),   // 36 / 37: 128 x 32
    SB2(2, 2, 2, 2, 1, 2, 2),   // 38 / 39: 128 x 64, eight waves
    // four loader waves beside the computing waves
    SB3(2, 1, 3, 1, 1, 3, 1, 4),   // 40 / 41:  64 x 32, 4 + 6 waves
    SB3(2, 1, 4, 1, 1, 2, 1, 4),   // 42 / 43:  64 x 32, 4 + 8 waves
    SB3(1, 1, 3, 1, 1, 3, 1, 4),   // 44 / 45:  32 x 32, 4 + 3 waves
    SB3(4, 1, 2, 1, 1, 2, 1, 4),   // 46 / 47: 128 x 32, 4 + 8 waves
    SB3(2, 2, 2, 1, 1, 2, 1, 4),   // 48 / 49:  64 x 64, 4 + 8 waves
    SB3(2, 2, 2, 2, 1, 2, 1, 4),   // 50 / 51: 128 x 64, 4 + 8 waves
    SB3(4, 1, 1, 1, 3, 4, 1, 4),   // 52 / 53: 128 x 96, 4 + 4 waves
    SB3(4, 2, 1, 1, 1, 4, 1, 4),   // 54 / 55: 128 x 64, 4 + 8 waves, no K sharing
    // eight loader waves
    SB3(2, 1, 3, 1, 1, 3, 1, 8),   // 56 / 57:  64 x 32, 8 + 6 waves
    SB3(2, 1, 2, 1, 1, 4, 1, 8),   // 58 / 59:  64 x 32, 8 + 4 waves
    SB3(1, 1, 3, 1, 1, 3, 1, 8),   // 60 / 61:  32 x 32, 8 + 3 waves
    SB3(4, 1, 2, 1, 1, 2, 1, 8),   // 62 / 63: 128 x 32, 8 + 8 waves
    SB3(4, 2, 1, 1, 1, 4, 1, 8),   // 64 / 65: 128 x 64, 8 + 8 waves
};
constexpr int kNumSbVariants = sizeof(kSbVariants) / sizeof(kSbVariants[0]);

// the geometry of a launch; returns the LDS bytes it needs, 0 when the variant cannot run the layer
int sb_geometry(const ConvArgs& a, const SbVariant& v, SbGeom& g) {
    g = SbGeom{};
    const int chunks = a.Cin / 32;
    g.taps = a.KH * a.KW;
    g.total_units = g.taps * chunks;
    const int budget = (160 * 1024) / v.wgs_per_cu - 1024;
    if (!v.gather) {
        g.a_rows = (v.bm + 2 * a.W + 2 + 15) / 16 * 16;
        g.units = 9;
        g.stage_bytes = g.a_rows * 64 + 9 * v.bn * 64;
        g.stages = chunks;
    } else {
        const int unit_bytes = (v.bm + v.bn) * 64;
        g.units = std::min(v.km * v.wk, g.total_units);
        // more than one stage: the ring holds at least two (fewer units per wave and stage where it would not)
        while (g.units > v.wk && g.units < g.total_units && 2 * g.units * unit_bytes > budget) g.units -= v.wk;
        g.stage_bytes = g.units * unit_bytes;
        g.stages = (g.total_units + g.units - 1) / g.units;
    }
    g.ns = std::min(g.stages, budget / g.stage_bytes);
    if (g.ns < 1 || (g.ns < 2 && g.stages > 1)) return 0;
    g.mt = (a.M + v.bm - 1) / v.bm;
    g.nt = (a.Cout_pad + v.bn - 1) / v.bn;   // (a last tile of 16 channels: f32 views only, the stores beyond Cout_pad are dropped)
    // what the tiles of an XCD share: with the channel tile innermost an XCD reads its pixel rows once and every weight
    // row; with the pixel tile innermost the other way round -- the larger operand is the one to share
    const double in_bytes = (double)a.N * a.H * a.W * a.Cin * 2, wt_bytes = (double)a.Cout_pad * a.K * 2;
    g.n_inner = in_bytes >= wt_bytes ? 1 : 0;
    g.zero_off = g.ns * g.stage_bytes;
    g.ablate = std::getenv("RMR_SB_ABLATE") ? std::atoi(std::getenv("RMR_SB_ABLATE")) : 0;   // read by development builds only
    const int reduce_bytes = v.wk > 1 ? v.wk * (v.bm / 32) * (v.bn / 32) * 4096 : 0;   // aliases the ring
    return std::max(g.zero_off + 1024, reduce_bytes);
}

}  // namespace

int conv_sb_num_variants() { return kNumSbVariants; }
ConvTile conv_sb_tile(int id) { return ConvTile{kSbVariants[id].bm, kSbVariants[id].bn, 32}; }

bool conv_sb_supported(const ConvArgs& a, int variant) {
    if (a.KH != a.KW || (a.KH != 1 && a.KH != 3) || a.pad != a.KH / 2 || a.Cin % 32 || a.Cin < 32 || !a.wt_t32) return false;
    if (a.pre || a.in8 || a.out8 || a.split > 1) return false;
    // planar channel groups (a C2f's chunks as slabs): the gathered form of the 1x1 layers, 16-byte pieces inside one slab, f16
    // SiLU outputs (the wide epilogue), no shortcut, every view below 4 GB
    const bool slabbed = a.in_slab_c || a.out_slab_c;
    if (slabbed) {
        if (a.KH != 1 || a.stride != 1 || a.res || a.out32 || !a.act || a.in_slab_c % 8 || a.out_slab_c % 8) return false;
        if (a.in_slab_c && (a.Cin > 2040 || a.Cin % a.in_slab_c)) return false;
        if (a.out_slab_c && (((a.out_cs | a.out_co) & 7) || a.Cout_pad % a.out_slab_c)) return false;
        const double in_span = (a.in_slab_c ? (double)(a.Cin / a.in_slab_c - 1) * a.in_slab_stride : 0.0) + (double)a.N * a.H * a.W * a.in_cs * 2;
        const double out_span = (a.out_slab_c ? (double)(a.Cout_pad / a.out_slab_c - 1) * a.out_slab_stride : 0.0) + (double)a.M * a.out_cs * 2;
        if (in_span >= 3.9e9 || out_span >= 3.9e9) return false;
    }
    if (variant < 0) return true;
    if (variant >= kNumSbVariants) return false;
    const SbVariant& v = kSbVariants[variant];
    if (slabbed && !v.gather) return false;
    if (!v.gather && (a.KH != 3 || a.stride != 1 || a.Ho != a.H || a.Wo != a.W)) return false;
    if (a.Cout_pad % v.bn && !(a.out32 && a.Cout_pad % 16 == 0 && v.bn == 32)) return false;   // the class logits: 16 channels
    SbGeom g;
    const int lds = sb_geometry(a, v, g);
    if (lds <= 0 || lds > 160 * 1024 / v.wgs_per_cu) return false;
    return (long)g.mt * g.nt <= 16384;   // one tile per workgroup: small batches only
}

void launch_conv_sb(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int variant) {
    if (variant < 0 || variant >= kNumSbVariants) fail(RMR_ERR_INVALID_ARGUMENT, "conv_sb: variant %d out of range", variant);
    if (!conv_sb_supported(a, variant)) fail(RMR_ERR_LOGIC, "conv_sb: layer not supported by variant %d", variant);
    const SbVariant& v = kSbVariants[variant];
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 4 || a.out_co % 4) fail(RMR_ERR_LOGIC, "conv_sb: misaligned view");
    if (a.in_bytes == 0 || a.in_bytes > 0xf0000000ull || a.wt_t32_bytes == 0)
        fail(RMR_ERR_LOGIC, "conv_sb: buffer sizes not set or input view larger than 3.75 GiB");
    static std::once_flag once;
    std::call_once(once, [] {
        for (const SbVariant& d : kSbVariants) (void)hipFuncSetAttribute((const void*)d.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    SbGeom g;
    const int lds = sb_geometry(a, v, g);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d b%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, variant);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    v.kernel<<<g.mt * g.nt, v.threads, lds, stream>>>(a, g);
    RMR_HIP(hipGetLastError());
}

// ---- several independent layers in one launch -------------------------------------------------------------------------
size_t conv_sb_group_bytes(int n) { return (size_t)n * sizeof(SbProblem); }

bool conv_sb_group_supported(const ConvArgs* args, int n, int variant) {
    if (n < 2 || n > 8 || variant < 0 || variant >= kNumSbVariants) return false;
    long tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (!conv_sb_supported(args[i], variant)) return false;
        SbGeom g;
        sb_geometry(args[i], kSbVariants[variant], g);
        tiles += (long)g.mt * g.nt;
    }
    return tiles <= 16384;
}

// fills host_buf (conv_sb_group_bytes(n)) with the problem table of the launch; returns the number of tiles
int conv_sb_group_build(const ConvArgs* args, int n, int variant, void* host_buf) {
    if (!conv_sb_group_supported(args, n, variant)) fail(RMR_ERR_LOGIC, "conv_sb: group not supported by variant %d", variant);
    SbProblem* p = (SbProblem*)host_buf;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        p[i] = SbProblem{};
        p[i].a = args[i];
        sb_geometry(args[i], kSbVariants[variant], p[i].g);
        p[i].tile0 = tiles;
        tiles += p[i].g.mt * p[i].g.nt;
        if (args[i].in_cs % 8 || args[i].in_co % 8 || args[i].out_cs % 4 || args[i].out_co % 4) fail(RMR_ERR_LOGIC, "conv_sb: misaligned view");
    }
    return tiles;
}

void launch_conv_sb_group(DeviceCtx& ctx, hipStream_t stream, const ConvArgs* args, int n, const void* dev_probs, int variant) {
    if (!conv_sb_group_supported(args, n, variant)) fail(RMR_ERR_LOGIC, "conv_sb: group not supported by variant %d", variant);
    const SbVariant& v = kSbVariants[variant];
    static std::once_flag once;
    std::call_once(once, [] {
        for (const SbVariant& d : kSbVariants) (void)hipFuncSetAttribute((const void*)d.group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    int lds = 0, tiles = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
        SbGeom g;
        lds = std::max(lds, sb_geometry(args[i], v, g));
        tiles += g.mt * g.nt;
        const ConvArgs& a = args[i];
        flops += a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
        bytes += 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    }
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d group of %d (M%d N%d K%d k%d ...) b%d", args[0].N, n, args[0].M, args[0].Cout_pad, args[0].K, args[0].KH, variant);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    v.group_kernel<<<tiles, v.threads, lds, stream>>>((const SbProblem*)dev_probs, n, tiles);
    RMR_HIP(hipGetLastError());
}

}  // namespace rmr
