// conv_t32_common.h -- what the 32x32x16 convolution kernels (conv_t32.hip, conv_g32.hip) share: the LDS-DMA
// and wait wrappers, the constant-index loop, and the epilogue (bias, SiLU, shortcut, f16 / f32 stores).
#pragma once
#include <type_traits>

#include "conv_igemm.h"

namespace rmr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace t32 {

__device__ __forceinline__ float silu_t(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// LDS-DMA: 64 lanes x 16 bytes land at lds_addr + lane * 16; source = rsrc base + voff + soff
__device__ __forceinline__ void dma16s(u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

// the lower half of the wave only: lanes 0..31 move 512 bytes to lds_addr + lane * 16, lanes 32..63 sit the instruction out
// (two waves share one KiB block: a row of DMA slots that has half as many blocks as the workgroup has waves)
__device__ __forceinline__ void dma16s_lo(u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 exec_hi, 0\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 exec_hi, -1"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

// the same with the non-temporal hint (streamed once: do not displace what the other tiles re-read from L2)
__device__ __forceinline__ void dma16s_nt(u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
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

// f(integral_constant<0>) ... f(integral_constant<N-1>): a loop whose index is a constant expression
template <int K, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        static_for<K + 1, N>(f);
    }
}

// Epilogue of one tile: bias, SiLU, residual; a lane holds 4 x 4 consecutive channels of one pixel per fragment
// (pixel = lane & 31 of fragment row i, channels 8 g + 4 (lane >> 5) + 0..3 of fragment column j).
// EPI: 0 = results leave through v_permlane32_swap pairs as 16-byte stores (32 contiguous bytes per pixel),
//      1 = through a per-wave LDS stage at smem + stg_base as whole rows (NREP * 64 contiguous bytes per pixel).
//
// The wide path (f16 output, 8-channel aligned views, SiLU) is written around one fact of gfx9: loads and stores
// share vmcnt and retire in order, so a load issued behind a store cannot be waited for without waiting for the
// store's round trip to memory.  Round 2's first version fetched the bias per channel group between the stores of
// the groups: twelve store round trips in a row per tile, none of it overlapped with anything (measured: 110 of
// 388 us on the 96-channel layers, the same on zeros).  Here every load of the epilogue -- the wave's bias values
// and, with a shortcut, its residual pixels -- is issued BEFORE the first store, so there is one wait (which also
// covers the operand DMAs already in flight for the next tile) and after it only arithmetic and stores.  Stores and
// residual loads go through buffer resources that start at the wave's first pixel row and end at row M: rows past
// the end are dropped / read as zero by the bounds check, no branch, and views beyond 4 GiB stay addressable.
// BIAS_LDS: the layer's bias vector sits in LDS at smem + bias_off (Cout_pad floats, written once per kernel) and is read
// where it is used (lgkmcnt, no registers held); otherwise the wave's values are fetched into registers up front.
// SCALE (the fp8 kernel; needs BIAS_LDS): the accumulators are multiplied by one scale per output channel before the
// bias is added; the scales are Cout_pad floats in LDS right behind the bias vector.
// RS / row_off (conv_w1d): fragment row r of the tile is pixel RS * r + row_off (RS = 2: the rows are 2-pixel Winograd tiles and
// the call stores their even or odd pixels); EPI = 0 only.
// OUT8 (the fp8 plan; EPI = 0): 1 = the eight f16 values of a lane leave ONLY as e4m3 bytes in a.out8 (rows of a.out8_cs
// bytes; rounded from the f16 value a quantiser pass over the stored tensor would read) -- for a tensor whose only reader
// is an e4m3 layer: half the store bytes and no quantiser pass; 2 = as f16 AND as e4m3.  One more half exchange per
// 32-channel block makes the e4m3 store 16 bytes per lane.
template <int MREP, int NREP, int EPI, bool RES, bool BIAS_LDS, bool NOSTORE, bool SCALE = false, int AUX = 0, int RS = 1, int OUT8 = 0>
__device__ __forceinline__ void epilogue_wide(const ConvArgs& a, floatx16 (&acc)[MREP][NREP], unsigned char* smem, int stg_base, int bias_off,
                                              int m0, int n0, int wm, int wn, int lane, int row_off = 0) {
    static_assert(RS == 1 || EPI == 0, "strided rows leave through the lane-pair stores");
    static_assert(OUT8 == 0 || (EPI == 0 && RS == 1), "the e4m3 copy leaves through the lane-pair stores");
    static_assert(!SCALE || BIAS_LDS, "scales live in LDS");
    constexpr int STG_PITCH = NREP * 64 + 16;   // bytes per pixel row of the epilogue stage
    const int fr = lane & 31, kq = lane >> 5;
    const int cq = kq * 4;
    const int mw0 = __builtin_amdgcn_readfirstlane(RS * (m0 + wm * MREP * 32) + row_off);   // the wave's first pixel row
    const int nw0 = __builtin_amdgcn_readfirstlane(n0 + wn * NREP * 32);   // and first output channel
    const long rows_left = (long)a.M - mw0;
    const auto view_bytes = [&](int cs) {
        const long b = rows_left * cs * 2;
        return (unsigned)(b <= 0 ? 0 : b > 0xfffffff0l ? 0xfffffff0l : b);
    };
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)((_Float16*)a.out + (long)mw0 * a.out_cs), 0, view_bytes(a.out_cs), 0x00020000);
    const unsigned out_lane = (unsigned)(RS * fr * a.out_cs + a.out_co + nw0 + kq * 8) * 2u;   // fragment row 0, channel group 0
    const long b8 = rows_left * (OUT8 ? a.out8_cs : 0);
    const __amdgpu_buffer_rsrc_t out8_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((unsigned char*)a.out8 + (OUT8 ? (long)mw0 * a.out8_cs : 0)), 0, (unsigned)(b8 <= 0 ? 0 : b8 > 0xfffffff0l ? 0xfffffff0l : b8), 0x00020000);
    const unsigned out8_lane = (unsigned)(fr * (OUT8 ? a.out8_cs : 0) + nw0 + kq * 16);
    // ---- every load of the epilogue, ahead of its first store ------------------------------------------------
    float4 bias[BIAS_LDS ? 1 : NREP][BIAS_LDS ? 1 : 4];
    if constexpr (!BIAS_LDS) {
#pragma unroll
        for (int j = 0; j < NREP; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias[j][g] = *(const float4*)(a.bias + nw0 + j * 32 + g * 8 + cq);
    }
    const int bias_lane = bias_off + (nw0 + cq) * 4;
    u32x4 rres[RES ? MREP : 1][RES ? NREP : 1][2];
    if constexpr (RES) {
        const __amdgpu_buffer_rsrc_t res_rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)a.res + (long)mw0 * a.res_cs), 0, view_bytes(a.res_cs), 0x00020000);
        const unsigned res_lane = (unsigned)(RS * fr * a.res_cs + a.res_co + nw0 + kq * 8) * 2u;
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp)
                    rres[i][j][gp] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, res_lane + (unsigned)(RS * i * 32 * a.res_cs + j * 32 + gp * 16) * 2u, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        // Channel groups gq and gq + 1 of a lane pair (l, l + 32) hold channels 8 gq + {0..3 | 4..7} and
        // 8 gq + 8 + {0..3 | 4..7}: one v_permlane32_swap per dword gives the lower lane all eight channels of
        // group gq and the upper lane those of group gq + 1.
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
            unsigned q8[2] = {0u, 0u};   // the e4m3 bytes of group pair 0, until group pair 1 completes the 32-channel block
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float v[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4 b;
                    if constexpr (BIAS_LDS)
                        b = *(const float4*)(smem + bias_lane + (j * 32 + (gp * 2 + h) * 8) * 4);
                    else
                        b = bias[j][gp * 2 + h];
                    const int r0 = (gp * 2 + h) * 4;
                    if constexpr (SCALE) {
                        const float4 sc = *(const float4*)(smem + bias_lane + a.Cout_pad * 4 + (j * 32 + (gp * 2 + h) * 8) * 4);
                        v[h * 4 + 0] = silu_t(acc[i][j][r0 + 0] * sc.x + b.x);
                        v[h * 4 + 1] = silu_t(acc[i][j][r0 + 1] * sc.y + b.y);
                        v[h * 4 + 2] = silu_t(acc[i][j][r0 + 2] * sc.z + b.z);
                        v[h * 4 + 3] = silu_t(acc[i][j][r0 + 3] * sc.w + b.w);
                    } else {
                        v[h * 4 + 0] = silu_t(acc[i][j][r0 + 0] + b.x);
                        v[h * 4 + 1] = silu_t(acc[i][j][r0 + 1] + b.y);
                        v[h * 4 + 2] = silu_t(acc[i][j][r0 + 2] + b.z);
                        v[h * 4 + 3] = silu_t(acc[i][j][r0 + 3] + b.w);
                    }
                }
                union {
                    u32x4 u;
                    _Float16 h[8];
                    unsigned w[4];
                } o;
                if constexpr (RES) {
                    // the shortcut is added in f32 before the one rounding, so the values are exchanged as f32
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[r]), __float_as_uint(v[4 + r]), false, false);
                        v[r] = __uint_as_float(sw[0]);
                        v[4 + r] = __uint_as_float(sw[1]);
                    }
                    // lower lane: v[0..3] own group gq, v[4..7] the upper lane's group gq; upper lane: v[0..3] the lower
                    // lane's group gq + 1, v[4..7] own -- in both cases its eight channels in order
                    union {
                        u32x4 u;
                        _Float16 h[8];
                    } rr;
                    rr.u = rres[i][j][gp];
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
                if constexpr (OUT8 != 0) {
                    const auto sat = [](_Float16 x) { return __builtin_amdgcn_fmed3f((float)x, -448.f, 448.f); };
                    int w0 = 0, w1 = 0;
                    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(o.h[0]), sat(o.h[1]), w0, false);
                    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(o.h[2]), sat(o.h[3]), w0, true);
                    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(o.h[4]), sat(o.h[5]), w1, false);
                    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(o.h[6]), sat(o.h[7]), w1, true);
                    if (gp == 0) {
                        q8[0] = (unsigned)w0, q8[1] = (unsigned)w1;
                    } else {
                        // lane pair (l, l + 32) holds channels [0..7 | 8..15] of group pair 0 and [16..23 | 24..31] of group pair
                        // 1: after the exchange the lower lane has bytes 0..15 and the upper lane bytes 16..31 of the block
                        const auto s0 = __builtin_amdgcn_permlane32_swap(q8[0], (unsigned)w0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(q8[1], (unsigned)w1, false, false);
                        const u32x4 q = kq ? u32x4{s0[0], s1[0], (unsigned)w0, (unsigned)w1} : u32x4{q8[0], q8[1], s0[1], s1[1]};
                        __builtin_amdgcn_raw_buffer_store_b128(q, out8_rsrc, out8_lane + (unsigned)(i * 32 * a.out8_cs + j * 32), 0, AUX);
                    }
                }
                if constexpr (OUT8 == 1) {
                    // no f16 copy: the tensor's only reader takes the e4m3 bytes
                } else if constexpr (EPI == 0) {
                    if constexpr (NOSTORE)
                        asm volatile("" : : "v"(o.u));
                    else
                        __builtin_amdgcn_raw_buffer_store_b128(o.u, out_rsrc, out_lane + (unsigned)(RS * i * 32 * a.out_cs + j * 32 + gp * 16) * 2u, 0, AUX);
                } else {
                    *(u32x4*)(smem + stg_base + fr * STG_PITCH + (j * 32 + gp * 16 + kq * 8) * 2) = o.u;
                }
            }
        }
        if constexpr (EPI == 1) {
            // the wave's 32 x (NREP * 32) block leaves as whole rows: NREP * 4 lanes per pixel
            constexpr int CPP = NREP * 4;   // 16-byte chunks per pixel
#pragma unroll
            for (int it = 0; it < (32 * CPP) / 64; ++it) {
                const int f = it * 64 + lane;
                const int px = f / CPP, ch = f % CPP;
                const u32x4 vv = *(const u32x4*)(smem + stg_base + px * STG_PITCH + ch * 16);
                if constexpr (NOSTORE)
                    asm volatile("" : : "v"(vv));
                else
                    __builtin_amdgcn_raw_buffer_store_b128(vv, out_rsrc, (unsigned)((i * 32 + px) * a.out_cs + a.out_co + nw0 + ch * 8) * 2u, 0, AUX);
            }
        }
    }
}

// bias_off: LDS offset of the bias vector (BIAS_LDS kernels); CAN_RES = false: the kernel's layers do not carry a
// shortcut in the network (1x1 and strided layers, the 256-channel head tiles), so the wide shortcut path and its
// registers are not compiled in -- a shortcut still works, through the 8-byte path below
template <int MREP, int NREP, int EPI, bool BIAS_LDS, bool CAN_RES, bool NOSTORE = false, int AUX = 0, int RS = 1>
__device__ __forceinline__ void epilogue(const ConvArgs& a, floatx16 (&acc)[MREP][NREP], unsigned char* smem, int stg_base, int bias_off,
                                         int m0, int n0, int wm, int wn, int lane, int row_off = 0) {
    const int fr = lane & 31, kq = lane >> 5;
    const int cq = kq * 4;
    // 16-byte stores need 8-channel alignment
    const bool wide = !a.out32 && a.act && ((a.out_cs | a.out_co) & 7) == 0 && (!a.res || (CAN_RES && ((a.res_cs | a.res_co) & 7) == 0));
    if (wide) {
        if constexpr (CAN_RES) {
            if (a.res) {
                epilogue_wide<MREP, NREP, EPI, true, BIAS_LDS, NOSTORE, false, AUX, RS>(a, acc, smem, stg_base, bias_off, m0, n0, wm, wn, lane, row_off);
                return;
            }
        }
        epilogue_wide<MREP, NREP, EPI, false, BIAS_LDS, NOSTORE, false, AUX, RS>(a, acc, smem, stg_base, bias_off, m0, n0, wm, wn, lane, row_off);
        return;
    }
    // everything else (f32 output, no activation, 4-channel aligned views): 8-byte pieces, loads as they come
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = RS * (m0 + (wm * MREP + i) * 32 + fr) + row_off;
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < NREP; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + (wn * NREP + j) * 32 + gq * 8 + cq;
                if (n >= a.Cout_pad) continue;   // conv_sb: a 32-channel tile over a 16-channel view (the class logits)
                const float4 b = *(const float4*)(a.bias + n);
                float v[4] = {acc[i][j][gq * 4 + 0] + b.x, acc[i][j][gq * 4 + 1] + b.y, acc[i][j][gq * 4 + 2] + b.z,
                              acc[i][j][gq * 4 + 3] + b.w};
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
    }
}

}  // namespace t32
}  // namespace rmr
