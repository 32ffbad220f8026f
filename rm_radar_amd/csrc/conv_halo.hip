// conv_halo.hip -- 3x3 / stride 1 / pad 1 convolution with the input staged ONCE per workgroup.
//
// conv_igemm / conv_dma gather the im2col matrix, i.e. every input pixel travels L2 -> LDS nine
// times (once per filter tap).  Measured on MI355X the K loop of those kernels is paced by the
// per-CU vector-memory path (~20 B/clk/CU for 64-128 B row segments): DMA issue takes 760 of the
// 1380 cycles a wave spends per K slice at 128x96.  For 3x3 convolutions -- 85 % of YOLOv8m's
// FLOPs -- the fix is to fetch fewer bytes per MAC:
//
//   * outputs are taken in LINEAR pixel order m = (n*H + y)*W + x (stride 1, so input and output
//     share that index).  A workgroup owns BM consecutive outputs; every tap of every one of them
//     lies in the linear input range [m0 - W - 1, m0 + BM + W + 1).  That range (BM + 2W + 2
//     pixels x 32 channels) is DMA'd into LDS once per 32-channel chunk, and tap (kh, kw) is just
//     a row shift of (kh-1)*W + (kw-1) when the MFMA fragments are read: 9 taps reuse one copy.
//     Taps that fall outside the image read a neighbouring row's pixel; they are zeroed in
//     registers from a per-pixel 9-bit validity mask (only border pixels have any).
//   * the LDS image is written lane-linearly by the DMA, so bank conflicts are handled on the
//     source side with the chunk permutation c ^ (((row >> 2) & 1) << 1), which is conflict-free
//     for ds_read_b128 fragment reads at EVERY row shift (exhaustively checked).
//   * weights still stream per (tap, chunk) slice through a 3-deep DMA ring with counted vmcnt;
//     the next chunk's input range is fetched in 8 portions alongside taps 0..7 of the current
//     one, so every slice issues the same number of DMA instructions (constant vmcnt immediates).
//
//   * TPS (taps per slice) = 1, 3 or 9 sets how much weight data one barrier-to-barrier slice
//     carries: a slice is TPS taps of one 32-channel chunk.  At TPS = 9 (narrow tiles, BN <= 96)
//     a 192-channel layer runs 6 slices instead of 54: that is what small batches need, where a
//     tile's K loop is a serial chain of per-slice costs and one workgroup per CU cannot hide it.
//
// Staged bytes per K slice at 256 x 192: 16 KiB instead of 28 KiB (im2col), and the A part no
// longer grows with the tile height.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include "conv_igemm.h"

namespace rmr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {


// v * rcp(1 + e^-v): the hardware reciprocal (1 ulp) instead of an IEEE division -- the epilogue's VALU
// work is not small beside a short K loop (48 values per lane per tile)
__device__ __forceinline__ float silu_h(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ void dma16h(u32x4 rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// chunk permutation key of an LDS row (see header)
__device__ __forceinline__ int hkey(int row) { return ((row >> 2) & 1) << 1; }

// A_SLOTS: input-range DMA instructions per slice (taps 0..7 carry them) => up to A_SLOTS*128 rows
template <int WM, int WN, int MREP, int NREP, int A_SLOTS, int TPS>
__global__ __launch_bounds__(WM* WN * 64) void conv_halo_kernel(const ConvArgs a, const int a_rows) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * MREP * 16;
    constexpr int BN = WN * NREP * 16;
    constexpr int NB = BN / 16;                        // weight DMA instructions per tap
    constexpr int SPC = 9 / TPS;                       // slices per 32-channel chunk
    constexpr int BSTAGES = TPS == 9 ? 2 : 3;          // weight-slice ring
    constexpr int SLOTS = A_SLOTS + NB * TPS;          // DMA instructions per slice (workgroup)
    constexpr int NI = (SLOTS + NW - 1) / NW;          // per wave
    constexpr int B_TAP_BYTES = BN * 64;
    constexpr int B_STAGE_BYTES = B_TAP_BYTES * TPS;
    static_assert(TPS == 1 || TPS == 3 || TPS == 9, "a slice is 1, 3 or 9 taps");
    static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves");
    static_assert((BSTAGES - 1) * NI <= 63, "vmcnt is 6 bits");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const int a_buf_bytes = a_rows * 64;              // one input-range buffer
    const int b_base = 2 * a_buf_bytes;               // weight ring behind the two input buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nt_count = a.Cout_pad / BN;
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = blockIdx.x & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int m0 = (lid / nt_count) * BM;
    const int n0 = (lid % nt_count) * BN;
    const int W = a.W;
    const int lo = m0 - W - 1;            // input pixel held by LDS row 0
    const int npix = a.N * a.H * a.W;     // == M

    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu),
                           sgpr(a.in_bytes), sgpr(0x00020000u)};
    const u32x4 wt_rsrc = {sgpr((unsigned)(size_t)a.wt), sgpr((unsigned)((size_t)a.wt >> 32) & 0xffffu),
                           sgpr(a.wt_bytes), sgpr(0x00020000u)};
    constexpr unsigned OOB = 0xfffffff0u;

    // ---- DMA bookkeeping of this lane -------------------------------------------------------
    const int lrow = lane >> 2;                       // row inside a 16-row DMA block
    const int lchunk = (lane & 3) ^ hkey(lrow);       // logical 16-byte chunk this lane fetches
    // a_rows is a multiple of 8 (>= 16); the last 16-row DMA block starts at a_rows - 16, overlapping its
    // predecessor by eight rows when a_rows is not a multiple of 16 (same data, same swizzle phase) instead
    // of running past the buffer
    const int na = (a_rows + 15) / 16;                // input-range DMA instructions per chunk
    const auto a_blk = [&](int ia) { return min(ia * 16, a_rows - 16); };  // first row of block ia
    // weight rows of this lane's B slots (slot q = wave + NW*j; q >= A_SLOTS are weights: instruction
    // wi = q - A_SLOTS covers rows 16 (wi % NB) .. +15 of tap wi / NB of the slice)
    int w_off[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int q = wave + NW * j;
        const int wi = q - A_SLOTS;
        const int row = (wi % NB) * 16 + lrow;
        w_off[j] = (q >= A_SLOTS && q < SLOTS) ? ((n0 + row) * a.Kp + (wi / NB) * a.Cin + lchunk * 8) * 2 : 0;
    }
    const int chunks = (a.Cin + 31) / 32;  // the last chunk may be partial (Cin % 32 != 0): the
    // missing channels are zero-filled on BOTH operands (a weight slice must not run into the next tap)
    const int ch_in_chunk = lchunk * 8;

    const unsigned scratch = sgpr(lds0 + b_base + BSTAGES * B_STAGE_BYTES);  // idle slots land here
    const int zero_off = b_base + BSTAGES * B_STAGE_BYTES;                   // 16 zero bytes: what padding taps read --
    // the head of the scratch KiB: idle slots are out-of-range loads, which write zeros, so it stays zero
    if (tid == 0) *(u32x4*)(smem + zero_off) = u32x4{0, 0, 0, 0};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // written before this wave reaches the first barrier

    // Issues the DMA instructions that ride on slice sl = cc * SPC + g: the weights of the slice
    // BSTAGES - 1 ahead and one SPC-th of the NEXT chunk's input range.
    // With BSTAGES dividing the slices of a chunk the ring slot of a slice is g % BSTAGES: a
    // compile-time constant at every call site of the unrolled kw loop when TPS == 1.
    auto issue = [&](int cc, int g, int kw) {
        const int ahead = g + BSTAGES - 1;                   // slice being fetched, relative to chunk cc
        const int ks_c = ahead >= SPC ? cc + 1 : cc, ks_g = ahead >= SPC ? ahead - SPC : ahead;
        const bool w_live = ks_c < chunks;
        const int wdelta = (ks_g * TPS * a.Cin + ks_c * 32) * 2;
        const int w_stage = TPS == 9 ? ((cc + 1) & 1) : TPS == 1 ? (kw + BSTAGES - 1) % BSTAGES : ahead % BSTAGES;
        const bool a_live = cc + 1 < chunks;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int q = wave + NW * j;  // wave-uniform slot
            if (q < A_SLOTS) {
                const int ia = g * A_SLOTS + q;     // DMA block within the input range
                const int r0 = a_blk(ia);
                const int p = lo + r0 + lrow;  // input pixel of this lane's row
                if (a_live && ia < na) {
                    unsigned off = OOB;
                    if (p >= 0 && p < npix && (cc + 1) * 32 + ch_in_chunk < a.Cin)
                        off = (unsigned)((p * a.in_cs + a.in_co + (cc + 1) * 32 + lchunk * 8) * 2);
                    dma16h(in_rsrc, sgpr(lds0 + ((cc + 1) & 1) * a_buf_bytes + r0 * 64), off);
                } else if (BSTAGES > 2) {
                    dma16h(in_rsrc, scratch, OOB);  // keeps the counted vmcnt immediates constant
                }
            } else if (q < SLOTS) {
                const unsigned off = (w_live && ks_c * 32 + ch_in_chunk < a.Cin) ? (unsigned)(w_off[j] + wdelta) : OOB;
                if (BSTAGES > 2 || w_live)  // a 2-deep ring drains to vmcnt(0): nothing to keep constant
                    dma16h(wt_rsrc, sgpr(lds0 + b_base + w_stage * B_STAGE_BYTES + (q - A_SLOTS) * 1024), off);
            } else if (BSTAGES > 2) {
                dma16h(wt_rsrc, scratch, OOB);
            }
        }
    };

    floatx4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: the whole input range of chunk 0, weight slices 0 .. BSTAGES-2 ------------
    for (int ia = wave; ia < na; ia += NW) {
        const int r0 = a_blk(ia);
        const int p = lo + r0 + lrow;
        unsigned off = OOB;
        if (p >= 0 && p < npix && ch_in_chunk < a.Cin) off = (unsigned)((p * a.in_cs + a.in_co + lchunk * 8) * 2);
        dma16h(in_rsrc, sgpr(lds0 + r0 * 64), off);
    }
    // weight slices 0 .. BSTAGES-2
#pragma unroll
    for (int s = 0; s < BSTAGES - 1; ++s) {
        const int s_c = s / SPC, s_g = s % SPC;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int q = wave + NW * j;
            if (q >= A_SLOTS && q < SLOTS)
                dma16h(wt_rsrc, sgpr(lds0 + b_base + s * B_STAGE_BYTES + (q - A_SLOTS) * 1024),
                       (s_c < chunks && s_c * 32 + ch_in_chunk < a.Cin) ? (unsigned)(w_off[j] + (s_g * TPS * a.Cin + s_c * 32) * 2) : OOB);
        }
    }
    wait_vm<0>();

    // ---- per-lane fragment bookkeeping ------------------------------------------------------
    const int frow = lane & 15;
    const int kg = lane >> 4;
    int a_row[MREP];         // LDS row of this lane's pixel (centre tap) per fragment
    unsigned a_mask[MREP];   // valid-tap bits of this lane's pixel per fragment
    // One division for the wave's first pixel, then 16-pixel steps (the setup runs once per tile and
    // a 27-slice K loop is short: two divisions and nine tests per fragment were a third of its VALU
    // work).  Valid taps = (rows kh with 0 <= y + kh - 1 < H) x (columns kw likewise): bit 3 kh + kw.
    {
        const int q0 = wm * MREP * 16 + frow;
        int x = (m0 + q0) % W;
        int y = ((m0 + q0) / W) % a.H;
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            a_row[i] = q0 + i * 16 + W + 1;
            const unsigned rows = (y > 0 ? 0x007u : 0u) | 0x038u | (y < a.H - 1 ? 0x1c0u : 0u);
            const unsigned cols = (x > 0 ? 0x049u : 0u) | 0x092u | (x < W - 1 ? 0x124u : 0u);
            a_mask[i] = (m0 + q0 + i * 16 < a.M) ? (rows & cols) : 0u;
            x += 16;
            while (x >= W) {  // W >= 1: a few steps at most for tiny maps
                x -= W;
                if (++y == a.H) y = 0;
            }
        }
    }
    const int b_frag = b_base + (wn * NREP * 16 + frow) * 64 + ((kg ^ hkey(frow)) * 16);

    int stage_off = 0;       // ring slot of the slice being computed
    for (int cc = 0; cc < chunks; ++cc) {
        const int a_buf = (cc & 1) * a_buf_bytes;
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int t = kh * 3 + kw;
            if (TPS == 1 || (TPS == 3 && kw == 0) || (TPS == 9 && t == 0)) {  // head of a slice
                wait_vm<(BSTAGES - 2) * NI>();
                __builtin_amdgcn_s_barrier();
                issue(cc, t / TPS, kw);
                stage_off = (TPS == 9 ? (cc & 1) : TPS == 3 ? kh : kw) * B_STAGE_BYTES;
            }
            const int shift = (kh - 1) * W + (kw - 1);
            const unsigned char* bp = smem + b_frag + stage_off + (t % TPS) * B_TAP_BYTES;
            half8 xf[MREP], wf[NREP];
#pragma unroll
            for (int j = 0; j < NREP; ++j) wf[j] = *(const half8*)(bp + j * 16 * 64);
            // a tap outside the image reads the zero block instead: ONE select on the address
            // before the read, not four on the data between the read and the MFMA
            const auto read_x = [&](int i) {
                const int row = a_row[i] + shift;
                const int at = a_buf + row * 64 + ((kg ^ hkey(row)) * 16);
                const bool ok = (a_mask[i] >> t) & 1u;
                return *(const half8*)(smem + (ok ? at : zero_off));
            };
            if (MREP > 4 && NREP <= 3) {
                // tall tiles: fragments two at a time, so that five of them are never live at once (the same
                // order costs the 256 x 128 tile 14 %, although it would take it from 129 to 117 VGPRs)
                xf[0] = read_x(0);
#pragma unroll
                for (int i = 0; i < MREP; ++i) {
                    if (i + 1 < MREP) xf[i + 1] = read_x(i + 1);
#pragma unroll
                    for (int j = 0; j < NREP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < MREP; ++i) xf[i] = read_x(i);
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < NREP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
          }
        }
    }
    wait_vm<0>();

    // ---- epilogue: bias, SiLU, residual, store 4 consecutive channels per lane ----
    const int px = lane & 15;
    const int cq = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = m0 + (wm * MREP + i) * 16 + px;
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            const int n = n0 + (wn * NREP + j) * 16 + cq;
            const float4 b = *(const float4*)(a.bias + n);
            float v[4] = {acc[i][j][0] + b.x, acc[i][j][1] + b.y, acc[i][j][2] + b.z, acc[i][j][3] + b.w};
            if (a.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu_h(v[r]);
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

struct HaloTile {
    int bm, bn, threads, max_rows, tps;
    void (*kernel)(const ConvArgs, int);
};

// A_SLOTS input-range DMA instructions ride on each of the 9 / TPS slices of a chunk
#define HTILE_T(WM, WN, MR, NR, AS, TPS) \
    { WM * MR * 16, WN * NR * 16, WM * WN * 64, AS * (9 / TPS) * 16, TPS, conv_halo_kernel<WM, WN, MR, NR, AS, TPS> }
#define HTILE(WM, WN, MR, NR) HTILE_T(WM, WN, MR, NR, 4, 1)
#define HTILE_WIDE(WM, WN, MR, NR) HTILE_T(WM, WN, MR, NR, 6, 1)

const HaloTile kHaloTiles[] = {
    HTILE(4, 2, 4, 6),  // 0: 256 x 192
    HTILE(4, 2, 4, 3),  // 1: 256 x 96
    HTILE(4, 2, 4, 9),  // 2: 256 x 288
    HTILE(4, 2, 4, 4),  // 3: 256 x 128
    HTILE(4, 2, 4, 8),  // 4: 256 x 256
    HTILE(4, 2, 4, 2),  // 5: 256 x 64
    HTILE(4, 2, 2, 6),  // 6: 128 x 192
    HTILE(4, 2, 2, 9),  // 7: 128 x 288
    HTILE(2, 2, 4, 3),  // 8: 128 x 96 (4 waves)
    HTILE(2, 2, 4, 4),  // 9: 128 x 128 (4 waves)
    HTILE(4, 1, 4, 6),  // 10: 256 x 96 (4 waves)
    // wide feature maps (W = 160: 322 halo rows) and N = 48: a longer input range per chunk
    HTILE_WIDE(4, 1, 4, 3),  // 11: 256 x 48 (4 waves)
    HTILE_WIDE(8, 1, 2, 3),  // 12: 256 x 48 (8 waves)
    HTILE_WIDE(4, 2, 4, 3),  // 13: 256 x 96 (8 waves)
    HTILE_WIDE(8, 1, 3, 3),  // 14: 384 x 48 (8 waves)
    // whole-chunk slices (TPS = 9): narrow tiles for small batches, one barrier per 32 channels
    HTILE_T(4, 1, 1, 3, 24, 9),  // 15:  64 x 48
    HTILE_T(4, 1, 2, 3, 24, 9),  // 16: 128 x 48
    HTILE_T(4, 1, 4, 3, 28, 9),  // 17: 256 x 48
    HTILE_T(4, 1, 1, 6, 24, 9),  // 18:  64 x 96
    HTILE_T(4, 1, 2, 6, 24, 9),  // 19: 128 x 96
    HTILE_T(4, 1, 1, 2, 24, 9),  // 20:  64 x 32
    HTILE_T(4, 1, 2, 2, 24, 9),  // 21: 128 x 32
    HTILE_T(4, 1, 1, 4, 24, 9),  // 22:  64 x 64
    HTILE_T(4, 1, 2, 4, 24, 9),  // 23: 128 x 64
    HTILE_T(2, 2, 1, 3, 24, 9),  // 24:  32 x 96
    // one barrier per filter row (TPS = 3)
    HTILE_T(4, 2, 2, 6, 6, 3),   // 25: 128 x 192
    HTILE_T(4, 2, 4, 6, 8, 3),   // 26: 256 x 192
    HTILE_T(4, 2, 4, 3, 10, 3),  // 27: 256 x 96
    // 16 waves: two 128-pixel halves share one weight ring (half the weight DMA per MAC at the
    // occupancy of two 8-wave workgroups)
    HTILE_T(8, 2, 2, 6, 4, 1),   // 28: 256 x 192
    HTILE_T(8, 2, 2, 3, 6, 1),   // 29: 256 x 96
    HTILE_T(8, 2, 2, 9, 4, 1),   // 30: 256 x 288
    HTILE_T(8, 2, 4, 3, 6, 1),   // 31: 512 x 96
    HTILE_T(8, 2, 4, 3, 16, 3),  // 32: 512 x 96, one barrier per filter row
    // 320 rows: at 256 images per launch the 40x40 maps give 1280 row blocks x 2 channel tiles = exactly
    // five rounds over 256 CUs x 2 workgroups, where 256-row tiles leave a quarter-full last round: 303 vs
    // 326 us on the 192-channel layers with a shortcut, 281 vs 293 without.  (Fragments are read two at a
    // time in these tiles: with all five live the 320 x 96 tile needs 138 VGPRs and loses its second
    // workgroup per CU.  On 80-wide maps its input range does not leave LDS room for two either.)
    HTILE(4, 2, 5, 6),           // 33: 320 x 192
    HTILE(4, 2, 5, 3),           // 34: 320 x 96
};
constexpr int kNumHaloTiles = sizeof(kHaloTiles) / sizeof(kHaloTiles[0]);

int halo_rows(int bm, int W) { return std::max(16, (bm + 2 * W + 2 + 7) / 8 * 8); }
int halo_lds_bytes(const HaloTile& t, int W) {
    const int stages = t.tps == 9 ? 2 : 3;
    return 2 * halo_rows(t.bm, W) * 64 + stages * t.bn * 64 * t.tps + 1024;  // + one scratch KiB for idle slots (its head is the zero block)
}

}  // namespace

int conv_halo_num_tiles() { return kNumHaloTiles; }
ConvTile conv_halo_tile(int id) { return ConvTile{kHaloTiles[id].bm, kHaloTiles[id].bn, 32}; }

bool conv_halo_supported(const ConvArgs& a, int tile) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin % 8 || a.Cin < 32) return false;
    if (a.Ho != a.H || a.Wo != a.W) return false;
    if (tile < 0) return true;
    const HaloTile& t = kHaloTiles[tile];
    return a.Cout_pad % t.bn == 0 && halo_rows(t.bm, a.W) <= t.max_rows && halo_lds_bytes(t, a.W) <= 160 * 1024;
}

void launch_conv_halo(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile) {
    if (tile < 0 || tile >= kNumHaloTiles) fail(RMR_ERR_INVALID_ARGUMENT, "conv_halo: tile %d out of range", tile);
    if (!conv_halo_supported(a, tile)) fail(RMR_ERR_LOGIC, "conv_halo: layer not supported by tile %d", tile);
    const HaloTile& t = kHaloTiles[tile];
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 4 || a.out_co % 4) fail(RMR_ERR_LOGIC, "conv_halo: misaligned view");
    if (a.in_bytes == 0 || a.in_bytes > 0xf0000000ull || a.wt_bytes == 0)
        fail(RMR_ERR_LOGIC, "conv_halo: buffer sizes not set or input view larger than 3.75 GiB");
    static std::once_flag once;
    std::call_once(once, [] {
        for (const HaloTile& d : kHaloTiles)
            (void)hipFuncSetAttribute((const void*)d.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int rows = halo_rows(t.bm, a.W);
    const int lds = halo_lds_bytes(t, a.W);
    const int grid = ((a.M + t.bm - 1) / t.bm) * (a.Cout_pad / t.bn);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d h%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, tile);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    t.kernel<<<grid, t.threads, lds, stream>>>(a, rows);
    RMR_HIP(hipGetLastError());
}

}  // namespace rmr
