// conv_dma.hip -- implicit-GEMM convolution with an LDS-DMA operand pipeline (gfx950).
//
// Same GEMM view, fragment layout and fused epilogue as conv_igemm.hip, different data path:
// both operand tiles go global -> LDS directly with `buffer_load_dwordx4 ... lds` (no staging
// VGPRs, no ds_write pass), into a ring of STAGES K-slices with a COUNTED `s_waitcnt vmcnt(N)`,
// so STAGES-1 slices are always in flight while the MFMAs of the oldest run: the K loop is no
// longer paced by one L2/HBM round trip per slice (the register-staged kernel measured
// SQ_WAIT_ANY = 44 % of wave cycles, MFMA pipe 22 % busy).
//
//  * zero fill: padding taps, rows past M and the K tail use a buffer offset past num_records;
//    the DMA then writes zeros into LDS -- no branch, no select.
//  * the DMA destination is wave-uniform base + lane*16, i.e. 16 rows x 64 B land contiguously,
//    so the bank-conflict fix is a swizzle on the SOURCE side: lane (row, p) fetches logical
//    16-byte chunk p ^ h(row >> 2), h = {0,2,3,1}, and fragment reads apply the same involution.
//    With that map the four 16-lane groups of a ds_read_b128 touch 16 distinct slots.
//  * requires Cin % 32 == 0 (a 32-wide K slice never straddles two filter taps, so the tap is
//    wave-uniform); the stem (Cin 8) and the Cin = 48 layers stay on conv_igemm.hip.
//  * one s_barrier per K slice; waits are hand-counted (the DMA is issued from inline asm, which
//    hipcc does not track, so it cannot drain the ring with a conservative vmcnt(0)).
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
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// 16 bytes per lane, global -> LDS.  m0 = wave-uniform LDS byte address; lane l lands at m0 + 16 l.
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// Source-side swizzle keys (involutions on the 16-byte chunk index) that make the four 16-lane
// groups of a ds_read_b128 fragment read hit 16 distinct LDS slots; `r` = row within the 16-row
// fragment.  BK = 32 (64-B rows): h = {0,2,3,1}[r >> 2];  BK = 64 (128-B rows): r >> 1.
template <int BK>
__device__ __forceinline__ int swz_key(int r) {
    return BK == 32 ? ((0x78 >> (2 * ((r >> 2) & 3))) & 3) : ((r >> 1) & 7);
}

template <int WM, int WN, int MREP, int NREP, int STAGES, int BK>
__global__ __launch_bounds__(WM* WN * 64) void conv_dma_kernel(const ConvArgs a) {
    constexpr int NW = WM * WN;  // waves per workgroup (4 or 8)
    constexpr int BM = WM * MREP * 16;
    constexpr int BN = WN * NREP * 16;
    constexpr int CPR = BK / 8;                 // 16-byte chunks per row of a K slice
    constexpr int RPD = 64 / CPR;               // rows covered by one 1-KiB DMA instruction
    constexpr int RB = BK * 2;                  // bytes per LDS row
    constexpr int ROWS = BM + BN;               // A rows then B rows of one K slice
    constexpr int NINST = ROWS / RPD;           // DMA instructions per slice
    constexpr int NI = (NINST + NW - 1) / NW;   // per wave (padded with no-op slots)
    constexpr int STAGE_BYTES = NI * NW * 1024; // incl. the pad slots
    constexpr int HALVES = BK / 32;             // 32-channel half-slices (each has ONE filter tap)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(BK == 32 || BK == 64, "BK is 32 or 64");
    static_assert(STAGES >= 2 && (STAGES - 1) * NI <= 63, "vmcnt is 6 bits");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nt_count = a.Cout_pad / BN;
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = blockIdx.x & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    // split-K: `split` consecutive logical ids (same XCD) share one output tile
    const int split = a.split > 1 ? a.split : 1;
    const int tile_id = lid / split;
    const int zsplit = lid - tile_id * split;
    const int m0 = (tile_id / nt_count) * BM;
    const int n0 = (tile_id % nt_count) * BN;

    // raw buffer descriptors {base lo, base hi (stride 0), num_records, flags}
    // (readfirstlane makes their uniformity provable, so they are allocated in SGPRs)
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu),
                           sgpr(a.in_bytes), sgpr(0x00020000u)};
    const u32x4 wt_rsrc = {sgpr((unsigned)(size_t)a.wt), sgpr((unsigned)((size_t)a.wt >> 32) & 0xffffu),
                           sgpr(a.wt_bytes), sgpr(0x00020000u)};

    // ---- what this lane fetches: instruction slot j of this wave covers rows RPD*(wave + NW*j) ..
    const int lrow = lane / CPR;  // row within the DMA block
    int s_off[NI];        // byte offset of (row, chunk) at K slice 0, tap (0,0)
    unsigned s_mask[NI];  // A rows: valid-tap bits; B rows: all ones; pad slots: 0
    int lhalf = 0;        // which 32-channel half of the slice this lane's chunk lies in
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int row = (wave + NW * j) * RPD + lrow;  // row in [A; B]
        // logical 16-byte chunk fetched into physical position (lane % CPR) of that row
        const int lchunk = (lane % CPR) ^ swz_key<BK>(row & 15);
        lhalf = lchunk >> 2;  // (rows of one lane share row & 15 parity pattern: see static_assert below)
        if (row < BM) {
            const int m = m0 + row;
            const bool ok = m < a.M;
            const int mm = ok ? m : 0;
            const int ow = mm % a.Wo;
            const int t = mm / a.Wo;
            const int oh = t % a.Ho;
            const int n = t / a.Ho;
            const int ih0 = oh * a.stride - a.pad;
            const int iw0 = ow * a.stride - a.pad;
            s_off[j] = (((n * a.H + ih0) * a.W + iw0) * a.in_cs + a.in_co + (lchunk & 3) * 8) * 2;
            unsigned mask = 0;
            if (ok) {
                for (int r = 0; r < a.KH; ++r)
                    for (int c = 0; c < a.KW; ++c)
                        if ((unsigned)(ih0 + r) < (unsigned)a.H && (unsigned)(iw0 + c) < (unsigned)a.W)
                            mask |= 1u << (r * a.KW + c);
            }
            s_mask[j] = mask;
        } else if (row < ROWS) {
            s_off[j] = ((n0 + row - BM) * a.Kp + lchunk * 8) * 2;
            s_mask[j] = 0xffffffffu;
        } else {
            s_off[j] = 0;
            s_mask[j] = 0;
        }
    }
    // every DMA block of one lane starts at a multiple of RPD * NW rows, a multiple of 16 for
    // both BK, so (row & 15) -- hence the swizzle key and lhalf -- is the same for all its slots
    static_assert((RPD * NW) % 16 == 0, "swizzle key must not depend on the slot");

    // K slices [kt_begin, kt_end) of this workgroup (all of them unless split-K)
    const int nk_all = (a.K + BK - 1) / BK;
    const int kt_begin = (int)((long)nk_all * zsplit / split);
    const int nk = (int)((long)nk_all * (zsplit + 1) / split);  // == kt_end
    // wave-uniform position of the next 32-channel half-slice in the filter window
    int k_ci, k_kw, k_kh;
    {
        const int k = kt_begin * BK;
        k_ci = k % a.Cin;
        const int t = k / a.Cin;
        k_kw = t % a.KW;
        k_kh = t / a.KW;
    }

    // slot kinds of this wave (wave-uniform, loop-invariant)
    unsigned k_a[NI];
    u32x4 k_rsrc[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const bool is_a = (wave + NW * j) * RPD < BM;
        k_a[j] = is_a ? 0xffffffffu : 0u;
        k_rsrc[j] = u32x4{is_a ? in_rsrc[0] : wt_rsrc[0], is_a ? in_rsrc[1] : wt_rsrc[1], is_a ? in_rsrc[2] : wt_rsrc[2],
                          in_rsrc[3]};
    }

    auto issue = [&](int kt, int stage) {
        const bool live = kt < nk;  // slots past the last slice are issued as no-ops (constant vmcnt)
        int delta[HALVES];
        unsigned tap_bit[HALVES];
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            delta[h] = ((k_kh * a.W + k_kw) * a.in_cs + k_ci) * 2;
            tap_bit[h] = (live && k_kh < a.KH) ? 1u << (k_kh * a.KW + k_kw) : 0u;
            k_ci += 32;
            if (k_ci >= a.Cin) {
                k_ci = 0;
                if (++k_kw == a.KW) {
                    k_kw = 0;
                    ++k_kh;
                }
            }
        }
        const int my_delta = HALVES == 2 && lhalf ? delta[HALVES - 1] : delta[0];
        const unsigned my_bit = HALVES == 2 && lhalf ? tap_bit[HALVES - 1] : tap_bit[0];
        const int wdelta = kt * BK * 2;
        const unsigned dead = live ? 0u : 0xffffffffu;  // wave-uniform
        // straight-line: both candidate offsets are one add each, the slot kind (k_a, loop-invariant)
        // picks one with mask arithmetic.  Written with branches this was ~100 scalar instructions and
        // a dozen jumps per slice -- 15 SALU per MFMA on the 256 x 96 tile.
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int blk = wave + NW * j;  // wave-uniform
            const unsigned off_a = (unsigned)(s_off[j] + my_delta) | ((s_mask[j] & my_bit) ? 0u : 0xffffffffu);
            const unsigned off_w = (unsigned)(s_off[j] + wdelta) | dead | (s_mask[j] ? 0u : 0xffffffffu);
            const unsigned off = (k_a[j] & off_a) | (~k_a[j] & off_w);  // all ones = out of range = zero fill / no-op
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + stage * STAGE_BYTES + blk * 1024));
            dma16(k_rsrc[j], dst, off);
        }
    };

    floatx4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // prologue: STAGES-1 slices in flight
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) issue(kt_begin + s, s);

    // fragment addressing: row (lane & 15) of a 16-row block, logical 16-byte chunk ks*4 + (lane >> 4)
    const int frow = lane & 15;
    const int fkey = swz_key<BK>(frow);
    const unsigned char* a_frag = smem + (wm * MREP * 16 + frow) * RB;
    const unsigned char* b_frag = smem + (BM + wn * NREP * 16 + frow) * RB;

    // debug timing (a.timing != null): cycles spent by one wave in each phase of the K loop
    // (compiled in only with -DRMR_CONV_TIMING_BUILD=1: the per-slice stamps and their selects cost
    // every launch ~15 scalar instructions per MFMA otherwise)
#if RMR_CONV_TIMING_BUILD
    const bool timed = a.timing != nullptr && blockIdx.x == gridDim.x / 2 && wave == 1;
#else
    constexpr bool timed = false;
#endif
    long long t_wait = 0, t_bar = 0, t_issue = 0, t_read = 0, t_mfma = 0;
    auto now = [&]() -> long long { return timed ? (long long)__builtin_readcyclecounter() : 0; };

    int stage = 0;
    for (int kt = kt_begin; kt < nk; ++kt) {
        const long long c0 = now();
        // slice kt has landed for this wave once at most STAGES-2 newer slices are outstanding
        wait_vmcnt<(STAGES - 2) * NI>();
        const long long c1 = now();
        __builtin_amdgcn_s_barrier();  // ... and for every wave; all are also done reading slice kt-1
        const long long c2 = now();
        {
            int nxt = stage + STAGES - 1;
            if (nxt >= STAGES) nxt -= STAGES;
            issue(kt + STAGES - 1, nxt);  // refills the buffer slice kt-1 was read from
        }
        const long long c3 = now();
        long long c4 = c3;
#pragma unroll
        for (int ks = 0; ks < HALVES; ++ks) {
            const int coff = (((ks * 4 + (lane >> 4)) ^ fkey) * 16) + stage * STAGE_BYTES;
            half8 xf[MREP], wf[NREP];
#pragma unroll
            for (int i = 0; i < MREP; ++i) xf[i] = *(const half8*)(a_frag + i * 16 * RB + coff);
#pragma unroll
            for (int j = 0; j < NREP; ++j) wf[j] = *(const half8*)(b_frag + j * 16 * RB + coff);
            if (timed) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                const long long t = now();
                t_read += t - c4;
                c4 = t;
            }
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            if (timed) {
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // let the last MFMA retire
                __builtin_amdgcn_sched_barrier(0);
                const long long t = now();
                t_mfma += t - c4;
                c4 = t;
            }
        }
        t_wait += c1 - c0, t_bar += c2 - c1, t_issue += c3 - c2;
        if (++stage == STAGES) stage = 0;
    }
    if (timed && lane == 0) {
        a.timing[0] = t_wait, a.timing[1] = t_bar, a.timing[2] = t_issue, a.timing[3] = t_read;
        a.timing[4] = t_mfma, a.timing[5] = nk;
    }
    wait_vmcnt<0>();  // the trailing no-op slots

    // ---- split-K: partial tiles meet in the workspace; the last arriver reduces them --------
    if (split > 1) {
        __shared__ int is_last;
        constexpr int TILE_F4 = NW * MREP * NREP * 64;  // float4 per partial tile
        float4* ws = (float4*)a.splitk_ws + ((size_t)tile_id * split + zsplit) * TILE_F4;
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
                ws[((wave * MREP + i) * NREP + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        // publish: every wave drains its stores, one lane releases at agent scope, then the ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ticket = __hip_atomic_fetch_add(a.splitk_cnt + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            is_last = ticket == split - 1;
            if (is_last) {
                __hip_atomic_store(a.splitk_cnt + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
        if (!is_last) return;
        // sum ALL partials (its own re-read from the workspace) in slice order: the result does not
        // depend on which workgroup happened to arrive last -- run-to-run deterministic
        const float4* wall = (const float4*)a.splitk_ws + (size_t)tile_id * split * TILE_F4;
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < split; ++z) {
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) {
                    const float4 p = wall[(size_t)z * TILE_F4 + ((wave * MREP + i) * NREP + j) * 64 + lane];
                    acc[i][j][0] += p.x, acc[i][j][1] += p.y, acc[i][j][2] += p.z, acc[i][j][3] += p.w;
                }
        }
    }

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
            float v[4] = {acc[i][j][0] + b.x, acc[i][j][1] + b.y, acc[i][j][2] + b.z, acc[i][j][3] + b.w};
            if (a.pre) {
                const float4 t = *(const float4*)(a.pre + mpre * a.pre_cs + n);
                v[0] += t.x, v[1] += t.y, v[2] += t.z, v[3] += t.w;
            }
            if (a.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
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

struct DmaTile {
    int bm, bn, bk, stages, threads, lds_bytes;
    void (*kernel)(const ConvArgs);
};

#define DTILE(WM, WN, MR, NR, ST, BK)                                                              \
    {                                                                                              \
        WM* MR * 16, WN* NR * 16, BK, ST, WM* WN * 64,                                             \
            ST*(((WM * MR * 16 + WN * NR * 16) / (512 / BK) + WM * WN - 1) / (WM * WN)) * (WM * WN) * 1024, \
            conv_dma_kernel<WM, WN, MR, NR, ST, BK>                                                \
    }

const DmaTile kDmaTiles[] = {
    DTILE(2, 2, 4, 3, 4, 32),  // 0: 128 x 96
    DTILE(4, 1, 4, 6, 3, 32),  // 1: 256 x 96
    DTILE(2, 2, 2, 3, 4, 32),  // 2:  64 x 96
    DTILE(2, 2, 4, 4, 4, 32),  // 3: 128 x 128
    DTILE(2, 2, 2, 4, 4, 32),  // 4:  64 x 128
    DTILE(2, 2, 4, 2, 4, 32),  // 5: 128 x 64
    DTILE(2, 2, 2, 2, 4, 32),  // 6:  64 x 64
    DTILE(4, 1, 4, 3, 4, 32),  // 7: 256 x 48
    DTILE(4, 1, 2, 3, 4, 32),  // 8: 128 x 48
    DTILE(4, 1, 1, 3, 4, 32),  // 9:  64 x 48
    DTILE(4, 1, 4, 1, 4, 32),  // 10: 256 x 16
    DTILE(4, 1, 1, 1, 4, 32),  // 11:  64 x 16
    DTILE(4, 1, 4, 2, 4, 32),  // 12: 256 x 32
    DTILE(4, 1, 1, 2, 4, 32),  // 13:  64 x 32
    // 8 waves: fewer staged bytes per MAC
    DTILE(4, 2, 4, 6, 3, 32),  // 14: 256 x 192
    DTILE(4, 2, 4, 3, 3, 32),  // 15: 256 x 96
    DTILE(8, 1, 4, 6, 3, 32),  // 16: 512 x 96
    DTILE(4, 2, 4, 4, 3, 32),  // 17: 256 x 128
    DTILE(4, 2, 4, 8, 3, 32),  // 18: 256 x 256
    DTILE(4, 2, 4, 9, 3, 32),  // 19: 256 x 288
    DTILE(4, 2, 2, 9, 3, 32),  // 20: 128 x 288
    DTILE(4, 2, 2, 6, 3, 32),  // 21: 128 x 192
    // BK = 64: every fetched row is a full 128-byte cache line (BK = 32 uses half of each line it
    // pulls from L2, and the K loop is paced by the vector-memory path), half the barriers
    DTILE(2, 2, 4, 3, 3, 64),  // 22: 128 x 96
    DTILE(4, 1, 4, 6, 2, 64),  // 23: 256 x 96
    DTILE(2, 2, 2, 3, 3, 64),  // 24:  64 x 96
    DTILE(2, 2, 4, 4, 3, 64),  // 25: 128 x 128
    DTILE(2, 2, 2, 4, 3, 64),  // 26:  64 x 128
    DTILE(2, 2, 4, 2, 3, 64),  // 27: 128 x 64
    DTILE(4, 1, 4, 3, 3, 64),  // 28: 256 x 48
    DTILE(4, 2, 4, 6, 2, 64),  // 29: 256 x 192 (8 waves)
    DTILE(4, 2, 4, 3, 3, 64),  // 30: 256 x 96  (8 waves)
    DTILE(4, 2, 4, 4, 2, 64),  // 31: 256 x 128 (8 waves)
    DTILE(4, 2, 2, 9, 2, 64),  // 32: 128 x 288 (8 waves)
    DTILE(4, 2, 2, 6, 3, 64),  // 33: 128 x 192 (8 waves)
    // deep rings for small grids (batch 1..4): the K loop of a lone workgroup per CU is paced by
    // DMA latency / (STAGES - 1), and LDS is free at that occupancy
    DTILE(2, 2, 2, 2, 8, 32),  // 34:  64 x 64, 8 stages
    DTILE(2, 2, 2, 3, 8, 32),  // 35:  64 x 96
    DTILE(4, 1, 1, 2, 8, 32),  // 36:  64 x 32
    DTILE(2, 2, 2, 4, 8, 32),  // 37:  64 x 128
    DTILE(2, 2, 4, 3, 6, 32),  // 38: 128 x 96
    DTILE(2, 2, 2, 2, 5, 64),  // 39:  64 x 64, BK 64
    DTILE(2, 2, 2, 3, 5, 64),  // 40:  64 x 96, BK 64
    DTILE(4, 1, 1, 2, 6, 64),  // 41:  64 x 32, BK 64
    DTILE(2, 2, 1, 2, 8, 32),  // 42:  32 x 64
    DTILE(2, 2, 1, 3, 8, 32),  // 43:  32 x 96
    // 320 rows: whole rounds at 256 images (1280 row blocks on the 40 x 40 maps), as in conv_halo
    DTILE(4, 2, 5, 6, 3, 32),  // 44: 320 x 192
    DTILE(4, 2, 5, 6, 2, 64),  // 45: 320 x 192, BK 64
    DTILE(4, 2, 5, 4, 3, 32),  // 46: 320 x 128
};
constexpr int kNumDmaTiles = sizeof(kDmaTiles) / sizeof(kDmaTiles[0]);

}  // namespace

bool conv_dma_supported(const ConvArgs& a) { return a.Cin % 32 == 0 && a.KH * a.KW <= 32; }

int conv_dma_pick_tile(int M, int cout_pad, int num_cus) {
    static const int want_bm = [] {
        const char* e = std::getenv("RMR_BM");
        return e ? std::atoi(e) : 0;
    }();
    static const int want_bk = [] {
        const char* e = std::getenv("RMR_BK");
        return e ? std::atoi(e) : 0;
    }();
    static const int want_bn = [] {
        const char* e = std::getenv("RMR_BN");
        return e ? std::atoi(e) : 0;
    }();
    // Largest tile area (fewest staged bytes per MAC) that still gives every CU >= 2 workgroups'
    // worth of waves; else the tile with the most workgroups.
    int best = -1, fallback = -1;
    long best_area = 0, fb_blocks = 0;
    for (int t = 0; t < kNumDmaTiles; ++t) {
        const DmaTile& d = kDmaTiles[t];
        if (cout_pad % d.bn) continue;
        if (want_bm && d.bm != want_bm) continue;
        if (want_bn && d.bn != want_bn) continue;
        if (want_bk && d.bk != want_bk) continue;
        const long blocks = (long)((M + d.bm - 1) / d.bm) * (cout_pad / d.bn);
        const long area = (long)d.bm * d.bn;
        const long need = d.threads == 512 ? num_cus : 2L * num_cus;
        if (blocks >= need && area > best_area) best = t, best_area = area;
        if (blocks > fb_blocks || (blocks == fb_blocks && fallback >= 0 && area > (long)kDmaTiles[fallback].bm * kDmaTiles[fallback].bn))
            fallback = t, fb_blocks = blocks;
    }
    return best >= 0 ? best : fallback;
}

void launch_conv_dma(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile) {
    if (tile < 0 || tile >= kNumDmaTiles) fail(RMR_ERR_INVALID_ARGUMENT, "conv_dma: tile %d out of range", tile);
    const DmaTile& t = kDmaTiles[tile];
    if (!conv_dma_supported(a)) fail(RMR_ERR_LOGIC, "conv_dma: needs Cin %% 32 == 0 (Cin = %d)", a.Cin);
    if (a.Cout_pad % t.bn) fail(RMR_ERR_LOGIC, "conv_dma: Cout_pad %d not a multiple of tile BN %d", a.Cout_pad, t.bn);
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 4 || a.out_co % 4)
        fail(RMR_ERR_LOGIC, "conv_dma: misaligned view");
    if (a.in_bytes == 0 || a.in_bytes > 0xf0000000ull || a.wt_bytes == 0)
        fail(RMR_ERR_LOGIC, "conv_dma: buffer sizes not set or input view larger than 3.75 GiB");
    static std::once_flag once;
    std::call_once(once, [] {
        for (const DmaTile& d : kDmaTiles)
            (void)hipFuncSetAttribute((const void*)d.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_bytes);
    });
    const int tiles = ((a.M + t.bm - 1) / t.bm) * (a.Cout_pad / t.bn);
    if (a.split > 1) {
        const int nk_all = (a.K + t.bk - 1) / t.bk;
        if (a.split > nk_all) a.split = nk_all;
        if (!a.splitk_ws || !a.splitk_cnt) fail(RMR_ERR_LOGIC, "conv_dma: split-K needs a workspace");
    }
    const int grid = tiles * (a.split > 1 ? a.split : 1);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d d%d/%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, tile, a.split > 1 ? a.split : 1);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    t.kernel<<<grid, t.threads, t.lds_bytes, stream>>>(a);
    RMR_HIP(hipGetLastError());
}

size_t conv_dma_splitk_ws_floats(const ConvArgs& a, int tile, int split) {
    const DmaTile& t = kDmaTiles[tile];
    const size_t tiles = (size_t)((a.M + t.bm - 1) / t.bm) * (a.Cout_pad / t.bn);
    return tiles * split * (size_t)t.bm * t.bn;
}
int conv_dma_splitk_tiles(const ConvArgs& a, int tile) {
    const DmaTile& t = kDmaTiles[tile];
    return ((a.M + t.bm - 1) / t.bm) * (a.Cout_pad / t.bn);
}

int conv_dma_num_tiles() { return kNumDmaTiles; }
ConvTile conv_dma_tile(int id) { return ConvTile{kDmaTiles[id].bm, kDmaTiles[id].bn, kDmaTiles[id].bk}; }

}  // namespace rmr
