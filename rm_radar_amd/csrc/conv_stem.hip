// conv_stem.hip -- the first layer of YOLOv8m: 3x3 / stride 2 / pad 1, 3 (stored as 8) -> 48
// channels, 640x640 -> 320x320.
//
// K is only 72 (9 taps x 8 stored channels) and the layer moves 16 B in and 24 B out per input
// pixel: it is an HBM stream, not a GEMM (1.05 GB per 64-image chunk, 0.13 ms at 8 TB/s).  The
// generic im2col kernel gathers every 16-byte tap separately and reaches 35 TFLOP/s (2.2 TB/s).
// Here a workgroup owns an 8 x 32 output tile:
//   * its 17 x 65 input patch (17.7 KB) is DMA'd into LDS once, lane-linear, out-of-image pixels
//     arriving as zeros from the buffer bounds check (that is the padding);
//   * a K step of the 16x16x32 MFMA is four taps: lane group kg reads tap 4 ks + kg of its
//     pixel, one ds_read_b128 (8 stored channels) per lane; three K steps cover the nine taps, the
//     three surplus tap slots meet zero weights;
//   * the 48 x 96 filter sits in 36 VGPRs for the whole kernel;
//   * results are staged in LDS and leave as 16-byte chunks, whole 96-byte pixels per six lanes.
//
// The LB instantiation has no f16 input at all: it fills the patch by sampling the BGR u8 source
// frames itself (the letterbox of preprocess.hip: resize + border + BGR->RGB + 1/255, same bytes),
// which removes the canvas write and read-back -- 13 of the 19 bytes per pixel that the pair of
// kernels moves besides the layer's own output.
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include "conv_igemm.h"
#include "preprocess.h"

namespace rmr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ST_TH = 8, ST_TW = 32;                  // output tile
constexpr int ST_PH = 2 * ST_TH + 1, ST_PW = 2 * ST_TW + 1;  // input patch (pixels)
constexpr int ST_PATCH = ST_PH * ST_PW;               // 1105 pixels of 16 B
constexpr int ST_DMA = (ST_PATCH + 63) / 64;          // 18 DMA instructions
constexpr int ST_NI = (ST_DMA + 3) / 4;               // per wave
constexpr int ST_PATCH_BYTES = ST_DMA * 1024;
constexpr int ST_OUT_BYTES = ST_TH * ST_TW * 96;
// round 4: the output stage ALIASES the patch (one more barrier per tile, after the last fragment read): 24.6 KB per
// workgroup instead of 43 KB, so five or six tiles are resident per CU where LDS allowed three (the kernel is VALU-bound
// at ~70 % VALU-busy with three waves per SIMD; registers now allow five)
constexpr int ST_LDS = ST_PATCH_BYTES > ST_OUT_BYTES ? ST_PATCH_BYTES : ST_OUT_BYTES;
constexpr int ST_LB_TABLES = (ST_PW + ST_PH) * 16;     // LB: column and row entries

// v * rcp(1 + e^-v): the hardware reciprocal (1 ulp) instead of an IEEE division -- the epilogue's VALU
// work is not small beside a short K loop (48 values per lane per tile)
__device__ __forceinline__ float silu_s(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ void dma16s(u32x4 rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc)
                 : "memory");
}

// (amdgpu_waves_per_eu(3): a register budget of 168 makes hipcc keep the accumulators in ArchVGPRs -- with the default budget
// of 512 it parks them in AccVGPRs and the epilogue pays one v_accvgpr_read per value, 48 per lane and tile, in a kernel
// that is VALU-bound: 68-72 % VALU-busy in profiles/r04_pmc_first_layers.txt)
template <bool LB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void conv_stem_kernel(const ConvArgs a, const LetterboxDesc* __restrict__ descs,
                                                        int fill, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, kg = lane >> 4;

    const int tiles_x = a.Wo / ST_TW, tiles_y = a.Ho / ST_TH;
    const int img = blockIdx.x / (tiles_x * tiles_y);
    const int t = blockIdx.x % (tiles_x * tiles_y);
    const int oy0 = (t / tiles_x) * ST_TH, ox0 = (t % tiles_x) * ST_TW;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;

    // ---- the filter: B fragments for 3 K steps x 3 channel tiles ------------------------------
    half8 wreg[3][3];
    const int cq = kg * 4;
    float4 bias[3];
    const auto load_filter = [&]() {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                wreg[ks][j] = *(const half8*)((const _Float16*)a.wt + (size_t)(j * 16 + frow) * a.Kp + ks * 32 + kg * 8);
#pragma unroll
        for (int j = 0; j < 3; ++j) bias[j] = *(const float4*)(a.bias + j * 16 + cq);
    };
    // LB: only after the patch is built -- 36 more live registers during the sampling passes cost a wave
    // per SIMD (204 vs 156 VGPRs), measured 0.319 vs 0.296 ms at 64 images
    if (!LB) load_filter();

    // ---- the input patch, lane-linear: LDS pixel id = row * ST_PW + col ------------------------
    if (LB) {
        // The bilinear geometry is separable: 65 column entries {byte offset, lx, kind, x_hi == x_lo}
        // and 17 row entries {offset of row y_lo, of row y_hi, ly, kind} are worked out once per tile
        // (the two IEEE divisions per pixel of letterbox_pixel() would otherwise be a third of this
        // kernel's instruction stream, which is what bounds it).  kind: 0 sample, 1 border, 2 outside
        // the canvas (the convolution's zero padding).
        const LetterboxDesc d = descs[img];
        const LetterboxSrc src = letterbox_src(d);
        uint4* const tcol = (uint4*)(smem + ST_LDS);
        uint4* const trow = tcol + ST_PW;
        if (tid < ST_PW) {
            const int ix = ix0 + tid, rx = ix - d.left;
            const bool in = rx >= 0 && rx < d.rw;
            const float src_x = (float)rx * (float)d.crop_w / (float)d.rw;
            const int x_lo = in ? (int)src_x : 0;
            uint4 e;
            e.x = in ? src.delta + (unsigned)(d.crop_x + x_lo) * 3u : 0u;
            e.y = in ? __float_as_uint(src_x - (float)x_lo) : 0u;
            e.z = (unsigned)ix >= (unsigned)a.W ? 2u : in ? 0u : 1u;
            e.w = x_lo + 1 > d.crop_w - 1;
            tcol[tid] = e;
        } else if (tid >= 128 && tid < 128 + ST_PH) {
            const int iy = iy0 + tid - 128, ry = iy - d.top;
            const bool in = ry >= 0 && ry < d.rh;
            const float src_y = (float)ry * (float)d.crop_h / (float)d.rh;
            const int y_lo = in ? (int)src_y : 0;
            const int y_hi = min(y_lo + 1, d.crop_h - 1);
            uint4 e;
            e.x = in ? (unsigned)(d.crop_y + y_lo) * (unsigned)d.src_stride : 0u;
            e.y = in ? (unsigned)(d.crop_y + y_hi) * (unsigned)d.src_stride : 0u;
            e.z = in ? __float_as_uint(src_y - (float)y_lo) : 0u;
            e.w = (unsigned)iy >= (unsigned)a.H ? 2u : in ? 0u : 1u;
            trow[tid - 128] = e;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the tables only: the filter loads stay in flight
        __builtin_amdgcn_s_barrier();
        unsigned char fpx[3] = {(unsigned char)fill, (unsigned char)fill, (unsigned char)fill};
        const uint4 fillv = letterbox_pixel_f16x8(fpx, scale);
        // 1105 pixels over 256 lanes: two waves take a fifth pass; which two alternates with the tile
        const int tid2 = (((wave + 2 * (int)(blockIdx.x & 1)) & 3) << 6) | lane;
        // all loads first (a pass whose pixel does not exist or is not sampled reads offset 0), one wait
        constexpr int NP = (ST_PATCH + 255) / 256;
        lb_u32x3 ta[NP], tb[NP];
        uint4 ec[NP], er[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int id = min(tid2 + 256 * k, ST_PATCH - 1);
            const int pr = id / ST_PW, pc = id - pr * ST_PW;
            ec[k] = tcol[pc];
            er[k] = trow[pr];
            ta[k] = letterbox_taps(src, er[k].x + ec[k].x);
            tb[k] = letterbox_taps(src, er[k].y + ec[k].x);
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {  // keeps the compiler from sinking the loads into the passes
            asm volatile("" : "+v"(ta[k]), "+v"(tb[k]));
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int id = tid2 + 256 * k;
            if ((id & ~63) >= ST_PATCH) break;  // wave-uniform
            unsigned char px[3];
            letterbox_blend(ta[k], tb[k], er[k].x + ec[k].x, er[k].y + ec[k].x, ec[k].w != 0, __uint_as_float(ec[k].y),
                            __uint_as_float(er[k].z), px);
            const unsigned kind = max(ec[k].z, er[k].w);
            uint4 v = letterbox_pixel_f16x8(px, scale);
            v = kind == 0 ? v : kind == 1 ? fillv : make_uint4(0, 0, 0, 0);
            if (id < ST_PATCH) *(uint4*)(smem + id * 16) = v;
        }
    } else {
        const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu),
                               sgpr(a.in_bytes), sgpr(0x00020000u)};
#pragma unroll
        for (int j = 0; j < ST_NI; ++j) {
            const int q = wave + 4 * j;  // wave-uniform
            if (q < ST_DMA) {
                const int id = q * 64 + lane;
                const int pr = id / ST_PW, pc = id - pr * ST_PW;
                const int iy = iy0 + pr, ix = ix0 + pc;
                const bool ok = id < ST_PATCH && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned off =
                    ok ? (unsigned)((((img * a.H + iy) * a.W + ix) * a.in_cs + a.in_co) * 2) : 0xffffffffu;
                dma16s(in_rsrc, sgpr(lds0 + q * 1024), off);
            }
        }
    }

    if (LB) load_filter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- wave w: output rows 2w, 2w + 1 of the tile, two 16-pixel fragments each ----------------
    // the accumulators start at the bias (round 4: 48 adds per lane and tile less in the epilogue; the f32 sum is bias + taps
    // instead of taps + bias, within the same half ulp of f32)
    floatx4 acc[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = floatx4{bias[j].x, bias[j].y, bias[j].z, bias[j].w};
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        int tap = 4 * ks + kg;
        tap = tap < 9 ? tap : 8;  // surplus slots: zero weights, any finite pixel
        const int kh = tap / 3, kw = tap - kh * 3;
        half8 xf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 2 * wave + (i >> 1), c = (i & 1) * 16 + frow;  // output pixel within the tile
            xf[i] = *(const half8*)(smem + ((2 * r + kh) * ST_PW + 2 * c + kw) * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[ks][j], xf[i], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue: bias + SiLU, f16, through LDS so that stores are whole pixels ----------------
    __syncthreads();   // every wave has read its last patch fragment: the stage may overwrite the patch
    unsigned char* const stage = smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 2 * wave + (i >> 1), c = (i & 1) * 16 + frow;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (a.act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_s(v[e]);
            }
            if (a.out32) {
                const long m = ((long)img * a.Ho + oy0 + r) * a.Wo + ox0 + c;
                *(float4*)(a.out32 + m * a.out_cs + a.out_co + j * 16 + cq) = make_float4(v[0], v[1], v[2], v[3]);
                continue;
            }
            union {
                uint2 u;
                _Float16 h[4];
            } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.h[e] = (_Float16)v[e];
            *(uint2*)(stage + (r * ST_TW + c) * 96 + (j * 16 + cq) * 2) = o.u;
        }
    }
    if (a.out32) return;
    // each wave staged its own two rows (LDS operations of a wave execute in order: no barrier); a tile
    // row is 192 chunks of 16 bytes, three per lane, at offsets that do not depend on the row
    asm volatile("" ::: "memory");
    unsigned coff[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int chunk = q * 64 + lane, c = chunk / 6, part = chunk - c * 6;
        coff[q] = (unsigned)((c * a.out_cs + a.out_co) * 2 + part * 16);
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        unsigned char* const rowp = (unsigned char*)a.out + (((long)img * a.Ho + oy0 + r) * a.Wo + ox0) * a.out_cs * 2;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            *(u32x4*)(rowp + coff[q]) = *(const u32x4*)(stage + (r * ST_TW * 6 + q * 64 + lane) * 16);
    }
}

}  // namespace

bool conv_stem_supported(const ConvArgs& a) {
    return a.KH == 3 && a.KW == 3 && a.stride == 2 && a.pad == 1 && a.Cin == 8 && a.Cout_pad == 48 && a.Kp >= 96 &&
           a.Wo % ST_TW == 0 && a.Ho % ST_TH == 0 && a.Ho == a.H / 2 && a.Wo == a.W / 2 && a.H % 2 == 0 && a.W % 2 == 0 &&
           !a.res && !a.pre;
}

static void launch_stem(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, const LetterboxDesc* descs, int fill,
                        float scale) {
    if (!conv_stem_supported(a)) fail(RMR_ERR_LOGIC, "conv_stem: layer not supported");
    if (a.out_cs % 8 || a.out_co % 8) fail(RMR_ERR_LOGIC, "conv_stem: misaligned view");
    if (!descs) {
        if (a.in_cs % 8 || a.in_co % 8) fail(RMR_ERR_LOGIC, "conv_stem: misaligned view");
        if (a.in_bytes == 0 || a.in_bytes > 0xf0000000ull) fail(RMR_ERR_LOGIC, "conv_stem: input view of 0 or more than 3.75 GiB");
    }
    const int grid = a.N * (a.Ho / ST_TH) * (a.Wo / ST_TW);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    // LB: 3 source bytes per canvas pixel at scale 1 (fewer when the source is magnified)
    const double in_bytes = descs ? 3.0 * a.N * a.H * a.W : 2.0 * a.N * a.H * a.W * a.Cin;
    const double bytes = in_bytes + 2.0 * ((double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[56];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d stem%s", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride,
                 descs ? "+letterbox" : "");
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    if (descs)
        conv_stem_kernel<true><<<grid, 256, ST_LDS + ST_LB_TABLES, stream>>>(a, descs, fill, scale);
    else
        conv_stem_kernel<false><<<grid, 256, ST_LDS, stream>>>(a, nullptr, 0, 0.f);
    RMR_HIP(hipGetLastError());
}

void launch_conv_stem(DeviceCtx& ctx, hipStream_t stream, ConvArgs a) { launch_stem(ctx, stream, a, nullptr, 0, 0.f); }

// descs: DEVICE array of a.N descriptors; a.in is not read, a.H x a.W is the canvas
void launch_conv_stem_letterbox(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, const LetterboxDesc* descs, int fill,
                                float scale) {
    if (!descs) fail(RMR_ERR_LOGIC, "conv_stem: no letterbox descriptors");
    launch_stem(ctx, stream, a, descs, fill, scale);
}

}  // namespace rmr
