// conv_pw.hip -- pointwise (1x1 / stride 1) convolution with the weights stationary in registers and a
// persistent walk over the pixels.
//
// The 1x1 layers of the 160x160 .. 40x40 levels have K = 96..768 and N = 96..384: 2-24 K slices per
// output tile.  In the tiled kernels (conv_dma) such a tile is all prologue and epilogue -- fill the
// ring, a handful of MFMAs, drain -- and every tile pulls the whole 18-147 KB weight matrix through
// the per-CU load path again (27 % of a 256 x 96 tile's bytes at K = 96).  They run at 2.5-3.3 TB/s
// of algorithmic traffic where an HBM stream can do 4+.  Here instead:
//
//   * one workgroup per CU owns ALL output channels; wave (wm, wn) keeps the B fragments of its 48
//     channels for the whole K in registers (K / 32 x 3 fragments = 36..144 VGPRs), loaded once;
//   * the workgroup walks row blocks of BM pixels (block b, b + G, b + 2G, ...): the input is one
//     continuous stream of stages, a stage = BM rows x 96 channels, DMA'd (buffer_load ... lds) into a
//     ring STAGES deep with counted vmcnt waits, so the next blocks' reads are in flight under the
//     MFMAs and the epilogue of the current one -- no per-tile pipeline fill;
//   * a stage is three sub-slices of 64-byte rows in conv_dma's layout (source-side chunk swizzle
//     {0,2,3,1}[row >> 2]: conflict-free ds_read_b128 fragment reads); one barrier per stage,
//     36 x MREP / 4 MFMAs per wave between barriers, A fragments are the only LDS reads;
//   * the epilogue (bias, optional half-resolution addend, SiLU, f16) runs every K / 96 stages and leaves through bounds-checked buffer
//     stores, which are always issued (rows past M get an out-of-range offset): loads and stores
//     retire in order on gfx9, so the stores of recent epilogues are simply part of the counted wait.
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

constexpr int PW_SPS = 3;  // K steps of 32 per stage
constexpr int PW_NREP = 3; // 16-channel tiles per wave

__device__ __forceinline__ float silu_p(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ void dma16p(u32x4 rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmp() {
    static_assert(N >= 0 && N <= 63, "vmcnt is 6 bits");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// epilogues among the STAGES - 1 stages before the one at position p of a block of E stages
constexpr int pw_recent_epilogues(int p, int E, int stages) {
    int c = 0;
    for (int k = 1; k <= stages - 1; ++k)
        if ((((p - k) % E) + E) % E == E - 1) ++c;
    return c;
}

// MODE 0: f16 output; 1: f32 output; 2: f16 output with the half-resolution f32 addend (ConvArgs::pre)
template <int KS, int WM, int WN, int MREP, int STAGES, int MODE>
__global__ __launch_bounds__(WM* WN * 64) void conv_pw_kernel(const ConvArgs a, const int n_blocks, const int n_tiles) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * MREP * 16;
    constexpr int E = KS / PW_SPS;            // stages per row block
    constexpr int SSB = BM * 64;              // bytes of one sub-slice (BM rows x 32 channels)
    constexpr int STAGE_BYTES = PW_SPS * SSB;
    constexpr int NINST = STAGE_BYTES / 1024; // DMA instructions per stage
    constexpr int NI = NINST / NW;            // per wave
    constexpr bool OUT32 = MODE == 1, PRE = MODE == 2;
    // stores per wave per epilogue: f32 results leave as one 16-byte store per 16 x 16 tile; f16 results are first
    // exchanged between lane rows (v_permlane16_swap) so that a lane holds 8 consecutive channels: two tiles per store
    constexpr int NST = OUT32 ? MREP * PW_NREP : MREP + MREP / 2;
    static_assert(MREP % 2 == 0 && PW_NREP == 3, "the f16 store pairing assumes three channel tiles and an even MREP");
    static_assert(KS % PW_SPS == 0 && NINST % NW == 0, "stage geometry");
    static_assert(STAGES >= 2 && STAGES <= 4 && E <= 8, "ring depth / stages per block");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, kg = lane >> 4;
    const int fkey = (0x78 >> (2 * ((frow >> 2) & 3))) & 3;  // conv_dma's 64-byte-row swizzle key

    // n_tiles workgroups (channel tiles of WN x 48) walk the same row blocks side by side; ids are dealt
    // round-robin to the 8 XCDs, so the tiles of one walker are 8 ids apart: same XCD, same L2
    const int G = gridDim.x / n_tiles;  // walkers (a multiple of 8 unless there is only one channel tile)
    const int nt = n_tiles > 1 ? (int)(blockIdx.x >> 3) % n_tiles : 0;
    const int w = n_tiles > 1 ? (int)((blockIdx.x >> 3) / n_tiles) * 8 + (int)(blockIdx.x & 7) : (int)blockIdx.x;
    const int nbase = nt * (WN * 48) + wn * 48;  // this wave's first output channel
    const int nb = w < n_blocks ? (n_blocks - w + G - 1) / G : 0;  // row blocks w, w + G, ...

    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu),
                           sgpr(a.in_bytes), sgpr(0x00020000u)};
    const unsigned scratch = sgpr(lds0 + STAGES * STAGE_BYTES);
    void* const outp = OUT32 ? (void*)a.out32 : (void*)a.out;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, 0xfffffff0u, 0x00020000);
    const u32x4 pre_rsrc = {sgpr((unsigned)(size_t)a.pre), sgpr((unsigned)((size_t)a.pre >> 32) & 0xffffu),
                            sgpr(0xfffffff0u), sgpr(0x00020000u)};
    const int hw = a.Ho * a.Wo;
    const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)a.Wo;

    // ---- the filter: B fragments of this wave's 48 channels, all K -----------------------------
    half8 wreg[KS][PW_NREP];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < PW_NREP; ++j)
            wreg[ks][j] = *(const half8*)((const _Float16*)a.wt + (size_t)(nbase + j * 16 + frow) * a.Kp + ks * 32 + kg * 8);
    const int cq = kg * 4;
    float4 bias[PW_NREP];
#pragma unroll
    for (int j = 0; j < PW_NREP; ++j) bias[j] = *(const float4*)(a.bias + nbase + j * 16 + cq);
    // opaque from here on (no rematerialisation from memory inside the walk); this also waits for them,
    // so no compiler-tracked load is outstanding once the hand-counted DMA starts
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < PW_NREP; ++j) asm volatile("" : "+v"(wreg[ks][j]));
#pragma unroll
    for (int j = 0; j < PW_NREP; ++j) asm volatile("" : "+v"(bias[j].x), "+v"(bias[j].y), "+v"(bias[j].z), "+v"(bias[j].w));

    // ---- DMA bookkeeping: instruction q = wave + NW j of a stage -> sub-slice q / (BM/16), rows
    // 16 (q % (BM/16)) + lane / 4, physical chunk lane % 4 holds logical chunk (lane % 4) ^ key(row)
    int d_row[NI];
    unsigned d_off[E][NI], d_dst[NI];  // per position of the stage within its row block
    {
        const int r16 = lane >> 2;
        const int lchunk = (lane & 3) ^ ((0x78 >> (2 * ((r16 >> 2) & 3))) & 3);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int q = wave + NW * j;
            const int sub = q / (BM / 16), rblk = q % (BM / 16);
            d_row[j] = rblk * 16 + r16;
            d_dst[j] = (unsigned)(q * 1024);
#pragma unroll
            for (int pos = 0; pos < E; ++pos) {
                const int ch = pos * (PW_SPS * 32) + sub * 32 + lchunk * 8;
                const int slab = a.in_slab_c ? ch / a.in_slab_c : 0;
                const int within = ch - slab * a.in_slab_c;
                d_off[pos][j] = (unsigned)slab * a.in_slab_stride + (unsigned)(d_row[j] * a.in_cs + a.in_co + within) * 2u;
            }
        }
    }
    int i_blk = 0;  // row block of the next stage to issue
    // a stage's DMA instructions: wave-uniform set-up once (issue_begin), then one instruction per call (issue_one) so
    // that the walk can place them behind its MFMAs; issue() = all of them at once (the prologue)
    unsigned is_base = 0, is_dead = 0;
    int is_m0 = 0;
    auto issue_begin = [&]() {
        const bool live = i_blk < nb;  // wave-uniform; past the end: no-ops keep the counts constant
        is_m0 = (w + i_blk * G) * BM;
        is_base = (unsigned)(is_m0 * a.in_cs) * 2u;
        is_dead = live ? 0u : 0xffffffffu;
    };
    auto issue_one = [&](int slot, const int pos, const int j) {  // pos, j: compile-time at every call site
        const unsigned off = (d_off[pos][j] + is_base) | is_dead | (is_m0 + d_row[j] < a.M ? 0u : 0xffffffffu);
        dma16p(in_rsrc, sgpr(lds0 + slot * STAGE_BYTES + d_dst[j]), off);
    };
    auto issue_end = [&](const int pos) {
        if (pos == E - 1) ++i_blk;
    };
    auto issue = [&](int slot, const int pos) {
        issue_begin();
#pragma unroll
        for (int j = 0; j < NI; ++j) issue_one(slot, pos, j);
        issue_end(pos);
    };

    // prologue: STAGES - 1 stages in flight.  In steady state the stores of epilogue u sit right after
    // the DMA of stage u + STAGES - 1; stand-ins (no-op loads) take the place of the epilogues "before
    // the first stage", so that the counted waits below hold from the first stage on
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {
        issue(s, s % E);
        if ((((s - STAGES + 1) % E) + E) % E == E - 1) {
#pragma unroll
            for (int k = 0; k < NST; ++k) dma16p(in_rsrc, scratch, 0xffffffffu);
        }
    }

    const unsigned a_base = (unsigned)((wm * MREP * 16 + frow) * 64 + ((kg ^ fkey) * 16));
    // this wave's 48 output channels lie in one slab
    const int oslab = a.out_slab_c ? nbase / a.out_slab_c : 0;
    const unsigned obase = (unsigned)oslab * a.out_slab_stride + (unsigned)(a.out_co + nbase - oslab * a.out_slab_c + (lane >> 4) * 4) * (OUT32 ? 4u : 2u);
    const int px = frow;
    // output offsets of this lane's MREP x 3 pieces within a row block (elements)
    const unsigned elt = OUT32 ? 4u : 2u;

    floatx4 acc[MREP][PW_NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < PW_NREP; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    int slot = 0;
    for (int b = 0; b < nb; ++b) {
        const int m0 = (w + b * G) * BM;
#pragma unroll
        for (int p = 0; p < E; ++p) {
            // this wave's share of the stage has landed once only the newer DMAs and the stores of the
            // epilogues issued since may still be outstanding (in-order retirement)
            switch (p) {  // p is a constant after unrolling; the template argument must be one before
                case 0: wait_vmp<(STAGES - 2) * NI + NST * pw_recent_epilogues(0, E, STAGES)>(); break;
                case 1: wait_vmp<(STAGES - 2) * NI + NST * pw_recent_epilogues(1 % E, E, STAGES)>(); break;
                case 2: wait_vmp<(STAGES - 2) * NI + NST * pw_recent_epilogues(2 % E, E, STAGES)>(); break;
                default: wait_vmp<(STAGES - 2) * NI + NST * pw_recent_epilogues(3 % E, E, STAGES)>(); break;  // p >= 3 >= STAGES - 1: none
            }
            __builtin_amdgcn_s_barrier();  // every wave's share; and the previous stage is fully consumed
            floatx4 tpre[MREP][PW_NREP];
            if (PRE && p == E - 1) {
                // the block's addends, issued BEFORE this stage's DMA: the epilogue then waits for
                // everything but that newest DMA (in-order retirement), i.e. for data requested a
                // stage of MFMAs ago at the least
#pragma unroll
                for (int i = 0; i < MREP; ++i) {
                    const int m = m0 + (wm * MREP + i) * 16 + px;
                    // m -> (image, y, x) with float reciprocals and a one-step correction (m < 2^24)
                    int img = (int)((float)m * inv_hw);
                    img += (img + 1) * hw <= m ? 1 : 0;
                    img -= img * hw > m ? 1 : 0;
                    const int rem = m - img * hw;
                    int y = (int)((float)rem * inv_w);
                    y += (y + 1) * a.Wo <= rem ? 1 : 0;
                    y -= y * a.Wo > rem ? 1 : 0;
                    const int x = rem - y * a.Wo;
                    const int mpre = (img * (a.Ho >> 1) + (y >> 1)) * (a.Wo >> 1) + (x >> 1);
                    const unsigned rowoff = (unsigned)(mpre * a.pre_cs + nbase + cq) * 4u;
                    const unsigned dead = m < a.M ? 0u : 0xffffffffu;
#pragma unroll
                    for (int j = 0; j < PW_NREP; ++j)
                        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen"
                                     : "=&v"(tpre[i][j])
                                     : "v"((rowoff + (unsigned)(j * 64)) | dead), "s"(pre_rsrc)
                                     : "memory");
                }
            }
            // the next stage's DMA instructions ride behind the MFMAs of this stage's K steps, NI_PER at a time: a
            // vector-memory instruction costs its wave 60-180 issue cycles, and the variants with one wave per SIMD
            // have nobody else to feed the matrix pipe meanwhile (the order of DMAs, addend loads and stores -- what
            // the counted waits rely on -- is unchanged)
            int nxt = slot + STAGES - 1;
            if (nxt >= STAGES) nxt -= STAGES;
            issue_begin();
            constexpr int NI_PER = (NI + PW_SPS - 1) / PW_SPS;
            constexpr int NMF = MREP * PW_NREP, HALF = (NMF + 1) / 2;
            const unsigned char* const sp = smem + slot * STAGE_BYTES + a_base;
#pragma unroll
            for (int ss = 0; ss < PW_SPS; ++ss) {
                half8 xf[MREP];
#pragma unroll
                for (int i = 0; i < MREP; ++i) xf[i] = *(const half8*)(sp + ss * SSB + i * 16 * 64);
#pragma unroll
                for (int c = 0; c < HALF; ++c)
                    acc[c / PW_NREP][c % PW_NREP] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[p * PW_SPS + ss][c % PW_NREP], xf[c / PW_NREP], acc[c / PW_NREP][c % PW_NREP], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = ss * NI_PER; j < (ss + 1) * NI_PER && j < NI; ++j) issue_one(nxt, (p + STAGES - 1) % E, j);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = HALF; c < NMF; ++c)
                    acc[c / PW_NREP][c % PW_NREP] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[p * PW_SPS + ss][c % PW_NREP], xf[c / PW_NREP], acc[c / PW_NREP][c % PW_NREP], 0, 0, 0);
            }
            issue_end((p + STAGES - 1) % E);
            if (++slot == STAGES) slot = 0;
            if (p == E - 1) {
                // ---- epilogue of the row block: bias, SiLU, store 4 consecutive channels per lane ----
                if (PRE) {
                    wait_vmp<NI>();
#pragma unroll
                    for (int i = 0; i < MREP; ++i)
#pragma unroll
                        for (int j = 0; j < PW_NREP; ++j) asm volatile("" : "+v"(tpre[i][j]));  // uses stay below the wait
                }
                u32x2 packed[MREP][PW_NREP];
#pragma unroll
                for (int i = 0; i < MREP; ++i) {
                    const int m = m0 + (wm * MREP + i) * 16 + px;
                    const unsigned dead = m < a.M ? 0u : 0xffffffffu;
                    const unsigned rowoff = (unsigned)(m * a.out_cs) * elt + obase;
#pragma unroll
                    for (int j = 0; j < PW_NREP; ++j) {
                        float v[4] = {acc[i][j][0] + bias[j].x, acc[i][j][1] + bias[j].y, acc[i][j][2] + bias[j].z,
                                      acc[i][j][3] + bias[j].w};
                        if (PRE) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += tpre[i][j][e];
                        }
                        if (a.act) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = silu_p(v[e]);
                        }
                        if (OUT32) {
                            const unsigned off = (rowoff + (unsigned)(j * 16) * elt) | dead;
                            u32x4 o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                            __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, off, 0, 0);
                        } else {
                            union {
                                u32x2 u;
                                _Float16 h[4];
                            } o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o.h[e] = (_Float16)v[e];
                            packed[i][j] = o.u;
                        }
                        acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                if (!OUT32) {
                    // A lane of row r = lane / 16 holds channels 4 r .. 4 r + 3 of every 16-channel tile.  v_permlane16_swap
                    // exchanges the odd rows of its first operand with the even rows of its second: on tiles A and B it
                    // leaves rows 0 / 2 with channels 0..7 / 8..15 of A and rows 1 / 3 with those of B -- one 16-byte
                    // store per lane for two tiles (64 contiguous bytes per pixel where A and B are neighbours).
                    const int lrow = lane >> 4;
                    const auto swap2 = [](u32x2 A, u32x2 B) {
                        const auto lo = __builtin_amdgcn_permlane16_swap(A.x, B.x, false, false);
                        const auto hi = __builtin_amdgcn_permlane16_swap(A.y, B.y, false, false);
                        return u32x4{lo[0], hi[0], lo[1], hi[1]};
                    };
                    // byte offset of channel 0 of the wave's tile 0 (obase carries this lane's 4 r as well)
                    const unsigned obase0 = obase - (unsigned)(lrow * 4) * 2u;
#pragma unroll
                    for (int i = 0; i < MREP; ++i) {
                        // tiles 0 and 1 of pixel block i: rows 0 / 2 store into tile 0, rows 1 / 3 into tile 1
                        const int m = m0 + (wm * MREP + i) * 16 + px;
                        const unsigned dead = m < a.M ? 0u : 0xffffffffu;
                        const unsigned off = (unsigned)(m * a.out_cs) * 2u + obase0 + (unsigned)((lrow & 1) * 16 + (lrow >> 1) * 8) * 2u;
                        __builtin_amdgcn_raw_buffer_store_b128(swap2(packed[i][0], packed[i][1]), out_rsrc, off | dead, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < MREP; i += 2) {
                        // tile 2 of pixel blocks i and i + 1: rows 0 / 2 store block i's pixel, rows 1 / 3 block i + 1's
                        const int m = m0 + (wm * MREP + i + (lrow & 1)) * 16 + px;
                        const unsigned dead = m < a.M ? 0u : 0xffffffffu;
                        const unsigned off = (unsigned)(m * a.out_cs) * 2u + obase0 + (unsigned)(32 + (lrow >> 1) * 8) * 2u;
                        __builtin_amdgcn_raw_buffer_store_b128(swap2(packed[i][2], packed[i + 1][2]), out_rsrc, off | dead, 0, 0);
                    }
                }
            }
        }
    }
    wait_vmp<0>();  // the trailing no-op DMAs must not outlive the workgroup's LDS
}

struct PwVariant {
    int ks, n, bm, stages, threads, lds_bytes;
    void (*k16)(const ConvArgs, int, int);
    void (*k32)(const ConvArgs, int, int);
    void (*kpre)(const ConvArgs, int, int);  // null: the variant has no registers to spare for the addend
};

#define PWV(KS, WM, WN, MR, ST)                                                                  \
    {                                                                                            \
        KS, WN * 48, WM* MR * 16, ST, WM* WN * 64, ST * 3 * (WM * MR * 16) * 64 + 1024,          \
            conv_pw_kernel<KS, WM, WN, MR, ST, 0>, conv_pw_kernel<KS, WM, WN, MR, ST, 1>, nullptr \
    }
#define PWV_PRE(KS, WM, WN, MR, ST)                                                              \
    {                                                                                            \
        KS, WN * 48, WM* MR * 16, ST, WM* WN * 64, ST * 3 * (WM * MR * 16) * 64 + 1024,          \
            conv_pw_kernel<KS, WM, WN, MR, ST, 0>, conv_pw_kernel<KS, WM, WN, MR, ST, 1>,        \
            conv_pw_kernel<KS, WM, WN, MR, ST, 2>                                                \
    }

const PwVariant kPw[] = {
    PWV(3, 4, 2, 2, 4),   // 0: K  96, N  96, 128-row blocks
    PWV(3, 4, 2, 4, 3),   // 1: K  96, N  96, 256-row blocks
    PWV(6, 4, 2, 2, 4),   // 2: K 192, N  96, 128
    PWV(6, 4, 2, 4, 3),   // 3: K 192, N  96, 256
    PWV_PRE(6, 2, 4, 4, 4),   // 4: K 192, N 192, 128
    PWV_PRE(6, 2, 4, 4, 3),   // 5: K 192, N 192, 128, shallower ring (two workgroups per CU)
    PWV(12, 2, 4, 4, 4),  // 6: K 384, N 192, 128
    PWV(12, 2, 4, 4, 3),  // 7: K 384, N 192, 128, shallower ring
    PWV(3, 4, 2, 2, 3),   // 8: K  96, N  96, 128, shallower ring
    PWV(6, 4, 2, 2, 3),   // 9: K 192, N  96, 128, shallower ring
    PWV(18, 1, 4, 4, 4),  // 10: K 576, N 192, 64-row blocks, 4 waves (216 VGPRs of weights: one wave per SIMD)
    PWV_PRE(12, 1, 4, 4, 4),  // 11: K 384, N 192, the same layout
    PWV(24, 1, 4, 4, 4),  // 12: K 768, N 192 per workgroup (288 VGPRs of weights)
};
constexpr int kNumPw = sizeof(kPw) / sizeof(kPw[0]);

}  // namespace

int conv_pw_num_variants() { return kNumPw; }

bool conv_pw_supported(const ConvArgs& a, int variant) {
    if (a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.Ho != a.H || a.Wo != a.W) return false;
    if (a.res || (a.pre && (a.out32 || a.Ho % 2 || a.Wo % 2 || a.M >= (1 << 24))) || a.split > 1 || a.Cin != a.K || a.K % 96 || a.Kp < a.K || a.in_bytes == 0) return false;
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 4 || a.out_co % 4 || (!a.out && !a.out32)) return false;
    if (a.in_slab_c % 8 || a.out_slab_c % 48) return false;  // 16-byte chunks / a wave's 48 channels stay within a slab
    // 32-bit byte offsets into the views
    const double in_span = (a.in_slab_c ? (double)(a.K / a.in_slab_c - 1) * a.in_slab_stride : 0.0) + (double)a.M * a.in_cs * 2;
    const double out_span = (a.out_slab_c ? (double)(a.Cout_pad / a.out_slab_c - 1) * a.out_slab_stride : 0.0) +
                            (double)a.M * a.out_cs * (a.out32 ? 4 : 2);
    if (in_span >= 4.0e9 || out_span >= 4.0e9) return false;
    // a workgroup owns v.n channels; wider layers run 2-3 workgroups side by side on the same rows
    const auto fits = [&](const PwVariant& v) {
        return v.ks * 32 == a.K && a.Cout_pad % v.n == 0 && a.Cout_pad / v.n <= 3 && (a.Cout_pad == v.n || v.n == 192) &&
               (!a.pre || v.kpre);
    };
    if (variant < 0) {
        for (int v = 0; v < kNumPw; ++v)
            if (fits(kPw[v])) return true;
        return false;
    }
    return variant < kNumPw && fits(kPw[variant]);
}

void launch_conv_pw(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int variant) {
    if (!conv_pw_supported(a, variant)) fail(RMR_ERR_LOGIC, "conv_pw: variant %d cannot run this layer", variant);
    const PwVariant& v = kPw[variant];
    const int n_blocks = (a.M + v.bm - 1) / v.bm;
    // persistent: one workgroup per CU, two where the ring leaves room
    const int per_cu = v.lds_bytes <= 80 * 1024 ? 2 : 1;
    const int n_tiles = a.Cout_pad / v.n;
    int walkers = std::min(n_blocks, std::max(1, ctx.num_cus * per_cu / n_tiles));
    if (n_tiles > 1) walkers = (walkers + 7) / 8 * 8;  // the id -> (walker, tile) map deals whole octets
    const int grid = walkers * n_tiles;
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = 2.0 * ((double)a.M * a.Cin + (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k1 s1 p%d", a.N, a.M, a.Cout_pad, a.K, variant);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        for (const PwVariant& pv : kPw) {
            (void)hipFuncSetAttribute((const void*)pv.k16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)pv.k32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (pv.kpre) (void)hipFuncSetAttribute((const void*)pv.kpre, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
    });
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    if (a.pre)
        v.kpre<<<grid, v.threads, v.lds_bytes, stream>>>(a, n_blocks, n_tiles);
    else if (a.out32)
        v.k32<<<grid, v.threads, v.lds_bytes, stream>>>(a, n_blocks, n_tiles);
    else
        v.k16<<<grid, v.threads, v.lds_bytes, stream>>>(a, n_blocks, n_tiles);
    RMR_HIP(hipGetLastError());
}

}  // namespace rmr
