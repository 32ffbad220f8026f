// net_ops.hip -- see net_ops.h.  All three are HBM/L2-bound elementwise passes: 16-byte
// (8 x f16) accesses per lane, channels fastest so a wave touches contiguous lines.
#include "net_ops.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace rmr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ half8 hmax8(half8 a, half8 b) {
    half8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = a[i] > b[i] ? a[i] : b[i];
    return r;
}

// One workgroup = one image x 8 channels: the H x W x 8 slab (6.4 KB at 20 x 20) is staged in LDS
// and the three chained 5x5 max-pools run as separable row / column passes on it (max is
// separable and -inf padding just shrinks the window), so x is read once and every pass works
// out of LDS.  Replaces a 13 x 13 brute-force sweep per pixel (57 us -> a few us at batch 1).
constexpr int SPPF_LDS_HALF8 = 3200;  // half8 slots: 2 buffers x pixels x channel groups per workgroup

// G consecutive 8-channel groups per workgroup: G adjacent threads read G x 16 contiguous bytes of a pixel
__global__ __launch_bounds__(256) void sppf_pools_kernel(__half* __restrict__ buf, int N, int H,
                                                         int W, int cs, int co, int C, int G) {
    __shared__ __attribute__((aligned(16))) half8 lds[SPPF_LDS_HALF8];
    const int npx = H * W, cnt = npx * G;
    half8* const cur = lds;
    half8* const tmp = lds + cnt;
    const int wg_per_img = C / (8 * G);
    const int n = blockIdx.x / wg_per_img, g0 = (blockIdx.x % wg_per_img) * G;
    _Float16* base = (_Float16*)buf + ((long)n * npx) * cs + co + g0 * 8;
    for (int i = threadIdx.x; i < cnt; i += 256) cur[i] = *(const half8*)(base + (long)(i / G) * cs + (i % G) * 8);
    __syncthreads();
    for (int level = 1; level <= 3; ++level) {
        for (int i = threadIdx.x; i < cnt; i += 256) {  // row pass
            const int p = i / G, q = i - p * G, y = p / W, x = p - y * W;
            half8 m = cur[i];
            for (int d = -2; d <= 2; ++d) {
                const int xx = x + d;
                if (d != 0 && xx >= 0 && xx < W) m = hmax8(m, cur[(y * W + xx) * G + q]);
            }
            tmp[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += 256) {  // column pass
            const int p = i / G, q = i - p * G, y = p / W, x = p - y * W;
            half8 m = tmp[i];
            for (int d = -2; d <= 2; ++d) {
                const int yy = y + d;
                if (d != 0 && yy >= 0 && yy < H) m = hmax8(m, tmp[(yy * W + x) * G + q]);
            }
            cur[i] = m;
            *(half8*)(base + (long)p * cs + level * C + q * 8) = m;
        }
        __syncthreads();
    }
}

void launch_sppf_pools(DeviceCtx& ctx, hipStream_t s, __half* buf, int N, int H, int W, int cs,
                       int co, int C) {
    if (2 * H * W > SPPF_LDS_HALF8) fail(RMR_ERR_LOGIC, "sppf: %dx%d feature map exceeds the LDS slab", H, W);
    // as many channel groups per workgroup as LDS holds (4 at 20 x 20: 64-byte reads), dividing C / 8
    // -- as long as the grid still covers the chip twice (small batches keep one group per workgroup)
    int G = 1;
    for (int g = 2; g <= 8; ++g)
        if ((C / 8) % g == 0 && 2 * H * W * g <= SPPF_LDS_HALF8 && (long)N * (C / 8 / g) >= 2L * ctx.num_cus) G = g;
    const long total = (long)N * H * W * (C / 8);
    ProfScope ps(ctx.prof, s, "sppf_pools", 0, (double)total * 16 * 4);
    sppf_pools_kernel<<<N * (C / 8 / G), 256, 0, s>>>(buf, N, H, W, cs, co, C, G);
    RMR_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void upsample2x_kernel(const __half* __restrict__ src, int src_cs,
                                                         int src_co, __half* __restrict__ dst,
                                                         int dst_cs, int dst_co, int N, int H, int W,
                                                         int C) {
    const int cg = C / 8;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * 2 * H * 2 * W * cg;
    if (idx >= total) return;
    const int g = (int)(idx % cg);
    long t = idx / cg;
    const int x = (int)(t % (2 * W));
    t /= 2 * W;
    const int y = (int)(t % (2 * H));
    const int n = (int)(t / (2 * H));
    const uint4 v = *(const uint4*)((const _Float16*)src + (((long)n * H + (y >> 1)) * W + (x >> 1)) * src_cs + src_co + g * 8);
    *(uint4*)((_Float16*)dst + (((long)n * 2 * H + y) * 2 * W + x) * dst_cs + dst_co + g * 8) = v;
}

void launch_upsample2x(DeviceCtx& ctx, hipStream_t s, const __half* src, int src_cs, int src_co,
                       __half* dst, int dst_cs, int dst_co, int N, int H, int W, int C) {
    const long total = (long)N * 4 * H * W * (C / 8);
    ProfScope ps(ctx.prof, s, "upsample2x", 0, (double)total * 16 * 1.25);
    upsample2x_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(src, src_cs, src_co, dst, dst_cs, dst_co, N, H, W, C);
    RMR_HIP(hipGetLastError());
}

// One thread per anchor.  Reads 64 + nc f32 logits (the anchor's row is contiguous: 256 B),
// writes 4 + nc f32 into the [C][A] planes (coalesced across the anchors of a wave).
__global__ __launch_bounds__(256) void head_decode_kernel(const float* __restrict__ box,
                                                          const float* __restrict__ cls, int cls_cs,
                                                          int nc, float* __restrict__ out, int N,
                                                          int H, int W, int stride, int a_off,
                                                          int a_total) {
    const int hw = H * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * hw) return;
    const int n = (int)(idx / hw);
    const int a = (int)(idx % hw);
    const float* b = box + idx * 64;
    float dist[4];
#pragma unroll
    for (int side = 0; side < 4; ++side) {
        float v[16];
        float mx = -3.0e38f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = *(const float4*)(b + side * 16 + q * 4);
            v[q * 4 + 0] = t.x;
            v[q * 4 + 1] = t.y;
            v[q * 4 + 2] = t.z;
            v[q * 4 + 3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, v[i]);
        float se = 0.f, sw = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __expf(v[i] - mx);
            se += e;
            sw += e * (float)i;
        }
        dist[side] = sw / se;
    }
    const float ax = (float)(a % W) + 0.5f, ay = (float)(a / W) + 0.5f;
    const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
    const float st = (float)stride;
    float* o = out + (long)n * (4 + nc) * a_total + a_off + a;
    o[0] = (x1 + x2) * 0.5f * st;
    o[(long)a_total] = (y1 + y2) * 0.5f * st;
    o[(long)2 * a_total] = (x2 - x1) * st;
    o[(long)3 * a_total] = (y2 - y1) * st;
    const float* c = cls + idx * cls_cs;
    for (int j = 0; j < nc; ++j) o[(long)(4 + j) * a_total] = 1.0f / (1.0f + __expf(-c[j]));
}

// the same over up to three scales in ONE launch (blockIdx.y = scale): at batch 1-4 a scale is 2-25 workgroups and a launch
// is mostly its fixed cost (three launches of ~5.4 us per forward in the batch-1 trace)
struct HeadScales {
    const float* box[3];
    const float* cls[3];
    int H[3], W[3], stride[3], a_off[3];
};

__global__ __launch_bounds__(256) void head_decode3_kernel(const HeadScales hs, int cls_cs, int nc, float* __restrict__ out, int N, int a_total) {
    const int sc = blockIdx.y;
    const int H = hs.H[sc], W = hs.W[sc];
    const int hw = H * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * hw) return;
    const int n = (int)(idx / hw);
    const int a = (int)(idx % hw);
    const float* b = hs.box[sc] + idx * 64;
    float dist[4];
#pragma unroll
    for (int side = 0; side < 4; ++side) {
        float v[16];
        float mx = -3.0e38f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = *(const float4*)(b + side * 16 + q * 4);
            v[q * 4 + 0] = t.x;
            v[q * 4 + 1] = t.y;
            v[q * 4 + 2] = t.z;
            v[q * 4 + 3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, v[i]);
        float se = 0.f, sw = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __expf(v[i] - mx);
            se += e;
            sw += e * (float)i;
        }
        dist[side] = sw / se;
    }
    const float ax = (float)(a % W) + 0.5f, ay = (float)(a / W) + 0.5f;
    const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
    const float st = (float)hs.stride[sc];
    float* o = out + (long)n * (4 + nc) * a_total + hs.a_off[sc] + a;
    o[0] = (x1 + x2) * 0.5f * st;
    o[(long)a_total] = (y1 + y2) * 0.5f * st;
    o[(long)2 * a_total] = (x2 - x1) * st;
    o[(long)3 * a_total] = (y2 - y1) * st;
    const float* c = hs.cls[sc] + idx * cls_cs;
    for (int j = 0; j < nc; ++j) o[(long)(4 + j) * a_total] = 1.0f / (1.0f + __expf(-c[j]));
}

void launch_head_decode3(DeviceCtx& ctx, hipStream_t s, int scales, const float* const* box, const float* const* cls, int cls_cs, int nc,
                         float* out, int N, const int* H, const int* W, const int* stride, const int* a_off, int a_total) {
    if (scales < 1 || scales > 3) fail(RMR_ERR_LOGIC, "head_decode3: %d scales", scales);
    HeadScales hs{};
    long most = 0, total = 0;
    for (int i = 0; i < scales; ++i) {
        hs.box[i] = box[i], hs.cls[i] = cls[i], hs.H[i] = H[i], hs.W[i] = W[i], hs.stride[i] = stride[i], hs.a_off[i] = a_off[i];
        most = std::max(most, (long)N * H[i] * W[i]);
        total += (long)N * H[i] * W[i];
    }
    ProfScope ps(ctx.prof, s, "head_decode", 0, (double)total * (64 + cls_cs + 4 + nc) * 4);
    head_decode3_kernel<<<dim3((unsigned)((most + 255) / 256), scales), 256, 0, s>>>(hs, cls_cs, nc, out, N, a_total);
    RMR_HIP(hipGetLastError());
}

// ---- the fused form: last 1x1 convolutions + decode --------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));
struct HeadFusedArgs {
    HeadFusedScale s[3];
};

// sum / max over the four lanes that share an anchor (lanes a, a + 16, a + 32, a + 48)
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

// KC = class-branch channels.  A workgroup of four waves takes 256 consecutive anchors of one scale; a wave takes four tiles
// of 16.  MFMA operands swapped as everywhere in the engine (D = W . X): lane (a = lane & 15, kg = lane >> 4) ends up with
// logits 4 kg .. 4 kg + 3 of every 16-channel tile for anchor a -- four bins of each side of the box distribution, four classes.
template <int KC>
__global__ __launch_bounds__(256) void head_fused_kernel(const HeadFusedArgs hs, int nc, float* __restrict__ out, int N, int a_total) {
    const HeadFusedScale S = hs.s[blockIdx.y];
    const int hw = S.H * S.W;
    const long total = (long)N * hw;
    const long wg0 = (long)blockIdx.x * 256;
    if (wg0 >= total) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, kg = lane >> 4;
    // both filters as A fragments, for the life of the workgroup
    half8 wbf[4][2], wcf[KC / 32];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wbf[j][ks] = *(const half8*)((const _Float16*)S.wb + (size_t)(j * 16 + frow) * S.kp_b + ks * 32 + kg * 8);
#pragma unroll
    for (int ks = 0; ks < KC / 32; ++ks) wcf[ks] = *(const half8*)((const _Float16*)S.wc + (size_t)frow * S.kp_c + ks * 32 + kg * 8);
    float4 bbv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bbv[j] = *(const float4*)(S.bb + j * 16 + kg * 4);
    const float4 bcv = *(const float4*)(S.bc + kg * 4);
    const float st = (float)S.stride;

#pragma unroll 2
    for (int t = 0; t < 4; ++t) {
        const long idx0 = wg0 + (wave * 4 + t) * 16;
        if (idx0 >= total) break;   // wave-uniform
        const long idx = idx0 + frow;
        const bool live = idx < total;
        const long ic = live ? idx : total - 1;
        const _Float16* const pb = (const _Float16*)S.hb + ic * S.cs_b + S.co_b + kg * 8;
        const _Float16* const pc = (const _Float16*)S.hc + ic * S.cs_c + S.co_c + kg * 8;
        half8 xb[2], xc[KC / 32];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xb[ks] = *(const half8*)(pb + ks * 32);
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) xc[ks] = *(const half8*)(pc + ks * 32);
        floatx4 box[4], cls = {bcv.x, bcv.y, bcv.z, bcv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            box[j] = floatx4{bbv[j].x, bbv[j].y, bbv[j].z, bbv[j].w};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) box[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wbf[j][ks], xb[ks], box[j], 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) cls = __builtin_amdgcn_mfma_f32_16x16x32_f16(wcf[ks], xc[ks], cls, 0, 0, 0);
        // DFL: side j = channel tile j; this lane holds bins 4 kg .. 4 kg + 3 of it
        float dist[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mx = quad_max(fmaxf(fmaxf(box[j][0], box[j][1]), fmaxf(box[j][2], box[j][3])));
            float se = 0.f, sw = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ex = __expf(box[j][e] - mx);
                se += ex;
                sw += ex * (float)(kg * 4 + e);
            }
            dist[j] = quad_sum(sw) / quad_sum(se);
        }
        const int n = (int)(ic / hw), a = (int)(ic % hw);
        const float ax = (float)(a % S.W) + 0.5f, ay = (float)(a / S.W) + 0.5f;
        const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
        float* const o = out + (long)n * (4 + nc) * a_total + S.a_off + a;
        if (live) {
            // the four lanes of an anchor write one box component each (16 consecutive anchors per component: 64-byte runs)
            const float comp = kg == 0 ? (x1 + x2) * 0.5f * st : kg == 1 ? (y1 + y2) * 0.5f * st : kg == 2 ? (x2 - x1) * st : (y2 - y1) * st;
            o[(long)kg * a_total] = comp;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = kg * 4 + e;
                if (c < nc) o[(long)(4 + c) * a_total] = 1.0f / (1.0f + __expf(-cls[e]));
            }
        }
    }
}

bool head_fused_supported(int kb, int kc, int nc) {
    return kb == 64 && nc >= 1 && nc <= 16 && (kc == 64 || kc == 96 || kc == 128 || kc == 192 || kc == 256);
}

void launch_head_fused(DeviceCtx& ctx, hipStream_t s, int scales, const HeadFusedScale* sc, int nc, float* out, int N, int a_total) {
    if (scales < 1 || scales > 3) fail(RMR_ERR_LOGIC, "head_fused: %d scales", scales);
    HeadFusedArgs hs{};
    long most = 0, total = 0;
    double flops = 0;
    for (int i = 0; i < scales; ++i) {
        if (sc[i].kc != sc[0].kc || !head_fused_supported(64, sc[i].kc, nc) || sc[i].kp_b < 64 || sc[i].kp_c < sc[i].kc || (sc[i].cs_b | sc[i].co_b | sc[i].cs_c | sc[i].co_c) & 7)
            fail(RMR_ERR_LOGIC, "head_fused: scale %d does not fit the kernel (kc %d, pitches %d / %d)", i, sc[i].kc, sc[i].cs_b, sc[i].cs_c);
        hs.s[i] = sc[i];
        const long m = (long)N * sc[i].H * sc[i].W;
        most = std::max(most, m);
        total += m;
        flops += 2.0 * m * (64.0 * 64 + (double)nc * sc[i].kc);
    }
    // per-layer profile name in the convolutions' form (bench.py groups by the trailing tag: y = this kernel)
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%ld N%d K%d k1 s1 y0", N, total, 64 + nc, 64 + sc[0].kc);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, s, pname, flops, (double)total * ((64 + sc[0].kc) * 2.0 + (4 + nc) * 4.0));
    const dim3 grid((unsigned)((most + 255) / 256), scales);
    switch (sc[0].kc) {
        case 64: head_fused_kernel<64><<<grid, 256, 0, s>>>(hs, nc, out, N, a_total); break;
        case 96: head_fused_kernel<96><<<grid, 256, 0, s>>>(hs, nc, out, N, a_total); break;
        case 128: head_fused_kernel<128><<<grid, 256, 0, s>>>(hs, nc, out, N, a_total); break;
        case 192: head_fused_kernel<192><<<grid, 256, 0, s>>>(hs, nc, out, N, a_total); break;
        default: head_fused_kernel<256><<<grid, 256, 0, s>>>(hs, nc, out, N, a_total); break;
    }
    RMR_HIP(hipGetLastError());
}

void launch_head_decode(DeviceCtx& ctx, hipStream_t s, const float* box, const float* cls,
                        int cls_cs, int nc, float* out, int N, int H, int W, int stride,
                        int a_off, int a_total) {
    const long total = (long)N * H * W;
    ProfScope ps(ctx.prof, s, "head_decode", 0, (double)total * (64 + cls_cs + 4 + nc) * 4);
    head_decode_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(box, cls, cls_cs, nc, out, N, H, W, stride, a_off, a_total);
    RMR_HIP(hipGetLastError());
}

}  // namespace rmr
