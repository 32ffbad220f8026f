// The 3-op prototype of VERDICT r04 item 1 and its kill criterion: "one C2f bottleneck chain at 20 x 20 -- if one persistent
// launch is not >= 25 % faster than its three launches, record and stop".
//
// Three dependent 3x3 layers of the 20 x 20 level of YOLOv8m (288 -> 288 channels, M = 400, K = 2592: a bottleneck's two
// convolutions with its shortcut, then the next bottleneck's first), one image, run
//   (a) as three launches of conv_sb's own kernel body (sb_tile, conv_sb.hip) back to back on one stream, and
//   (b) as ONE launch that walks the three layers: the same sb_tile per (workgroup, layer), a grid barrier between layers
//       (flat counter, agent-scope release before the arrival, acquire after the wait, as MI355X_MICROARCH.md prescribes for
//       data that crosses XCDs: the next layer's input was written by other CUs behind other L2s).
// Both run the SAME tiles through the SAME code, so the outputs must be bit-identical (checked), and the difference in time is
// exactly the seam: two kernel boundaries against two grid barriers.  sb_tile is included from the product source, not
// copied.  Every instantiation here is one whose waves all pass the same number of workgroup barriers inside a tile (no loader
// waves together with a shared K range: a loader leaves sb_tile two barriers short of the others, which a second layer in the
// same launch cannot follow), among them the product's variants 0, 2 and 52; the three-launch time of the tuner's variants for
// this layer (loaders AND shared K) is printed beside them.
//
// build (from the repo root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rm_radar_amd/csrc -I include tools/microbench/sb_chain.hip \
//                             -L rm_radar_amd/_build -lrmr -Wl,-rpath,'$ORIGIN/../../rm_radar_amd/_build' -o tools/microbench/sb_chain
// run: tools/microbench/sb_chain            -> profiles/r05_chain_prototype.txt
#include "../../rm_radar_amd/csrc/conv_sb.hip"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace rmr {
namespace {

struct ChainSync {
    unsigned count;   // arrivals, monotonic over a launch
    unsigned pad[31];
    unsigned fail;
};

__device__ __forceinline__ void chain_barrier(ChainSync* s, unsigned phase, unsigned n_wg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (phase + 1) * n_wg;
        unsigned spins = 0;
        while (__hip_atomic_load(&s->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) {
                s->fail = 1;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// one launch, n_ops dependent layers: workgroup b runs tile b of every layer that has one (every workgroup joins every barrier)
template <int WM, int WN, int WK, int MREP, int NREP, int LW>
__global__ __launch_bounds__((WM * WN * WK + LW) * 64) void chain_kernel(const SbProblem* __restrict__ probs, const int n_ops, ChainSync* sync) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    for (int p = 0; p < n_ops; ++p) {
        const ConvArgs a = probs[p].a;
        const SbGeom g = probs[p].g;
        const int n_tiles = g.mt * g.nt;
        if ((int)blockIdx.x < (n_tiles + 7) / 8 * 8) {
            const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
            const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
            const int cnt = q8 + (xcd < r8 ? 1 : 0);
            if (k < cnt) {
                const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
                const int mi = g.n_inner ? lid / g.nt : lid % g.mt, ni = g.n_inner ? lid % g.nt : lid / g.mt;
                sb_tile<WM, WN, WK, MREP, NREP, false, (9 + WK - 1) / WK, LW>(a, g, mi * (WM * MREP * 32), ni * (WN * NREP * 32), smem, lds0);
            }
        }
        if (p + 1 < n_ops) chain_barrier(sync, (unsigned)p, gridDim.x);
    }
}

// the chain on ONE XCD.  Workgroup ids are dealt round-robin over the eight XCDs (tools/microbench/
// wg_placement.hip), so of a grid of 8 * P workgroups only those with (id & 7) == 0 stay -- P workgroups behind ONE L2 -- and
// each walks tiles j, j + P, ... of every layer.  What a layer publishes never has to leave that L2: the barrier's release is
// a workgroup-scope one (s_waitcnt: the vector L1 is write-through, the stores are in the L2), the arrival an L2 atomic, the
// acquire invalidates the vector L1 only -- no buffer_wbl2, the instruction that makes the chip-wide barrier cost 8-10 us.
// P must be co-resident: 32 CUs x one workgroup of this LDS size.
__device__ __forceinline__ void xcd_barrier(ChainSync* s, unsigned phase, unsigned n_wg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (phase + 1) * n_wg;
        unsigned spins = 0;
        while (__hip_atomic_load(&s->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) {
                s->fail = 1;
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv: every wave drops its CU's stale L1 lines
}

template <int WM, int WN, int WK, int MREP, int NREP, int LW>
__global__ __launch_bounds__((WM * WN * WK + LW) * 64) void chain_xcd_kernel(const SbProblem* __restrict__ probs, const int n_ops, ChainSync* sync, const int P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    if (blockIdx.x & 7) return;            // the other seven XCDs' workgroups leave at once
    const int j = blockIdx.x >> 3;         // 0 .. P-1, all on XCD 0
    for (int p = 0; p < n_ops; ++p) {
        const ConvArgs a = probs[p].a;
        const SbGeom g = probs[p].g;
        const int n_tiles = g.mt * g.nt;
        for (int lid = j; lid < n_tiles; lid += P) {
            const int mi = g.n_inner ? lid / g.nt : lid % g.mt, ni = g.n_inner ? lid % g.nt : lid / g.mt;
            sb_tile<WM, WN, WK, MREP, NREP, false, (9 + WK - 1) / WK, LW>(a, g, mi * (WM * MREP * 32), ni * (WN * NREP * 32), smem, lds0);
            __syncthreads();               // the tile's LDS is reused by the next one
        }
        if (p + 1 < n_ops) xcd_barrier(sync, (unsigned)p, (unsigned)P);
    }
}

// the same tile walk as one launch per layer (conv_sb_kernel's mapping needs a grid of exactly n_tiles; this one takes the
// chain's grid so that (a) and (b) differ in the seam only)
template <int WM, int WN, int WK, int MREP, int NREP, int LW>
__global__ __launch_bounds__((WM * WN * WK + LW) * 64) void one_kernel(const SbProblem* __restrict__ probs, const int p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const ConvArgs a = probs[p].a;
    const SbGeom g = probs[p].g;
    const int n_tiles = g.mt * g.nt;
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    if (k >= q8 + (xcd < r8 ? 1 : 0)) return;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    const int mi = g.n_inner ? lid / g.nt : lid % g.mt, ni = g.n_inner ? lid % g.nt : lid / g.mt;
    sb_tile<WM, WN, WK, MREP, NREP, false, (9 + WK - 1) / WK, LW>(a, g, mi * (WM * MREP * 32), ni * (WN * NREP * 32), smem, lds0);
}

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

struct Layer {
    ConvArgs a;
    __half *w, *w32;
    float* bias;
};

template <int WM, int WN, int WK, int MREP, int NREP, int LW>
void run_shape(const char* label, std::vector<Layer>& layers, __half* const* bufs, size_t buf_elems, int reps) {
    const int n_ops = (int)layers.size();
    SbVariant v{};
    v.bm = WM * MREP * 32, v.bn = WN * NREP * 32, v.wk = WK, v.km = (9 + WK - 1) / WK, v.threads = (WM * WN * WK + LW) * 64, v.wgs_per_cu = 1, v.loaders = LW, v.gather = false;
    std::vector<SbProblem> probs(n_ops);
    int lds = 0, grid = 0;
    for (int p = 0; p < n_ops; ++p) {
        probs[p] = SbProblem{};
        probs[p].a = layers[p].a;
        const int need = sb_geometry(layers[p].a, v, probs[p].g);
        if (need <= 0 || need > 160 * 1024) {
            std::printf("%-28s does not fit this layer\n", label);
            return;
        }
        lds = std::max(lds, need);
        grid = std::max(grid, (probs[p].g.mt * probs[p].g.nt + 7) / 8 * 8);
    }
    SbProblem* dprobs;
    ChainSync* sync;
    CK(hipMalloc(&dprobs, sizeof(SbProblem) * n_ops));
    CK(hipMalloc(&sync, sizeof(ChainSync)));
    CK(hipMemcpy(dprobs, probs.data(), sizeof(SbProblem) * n_ops, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)chain_kernel<WM, WN, WK, MREP, NREP, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)one_kernel<WM, WN, WK, MREP, NREP, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<__half> out_a(buf_elems), out_b(buf_elems);
    CK(hipFuncSetAttribute((const void*)chain_xcd_kernel<WM, WN, WK, MREP, NREP, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<__half> out_c(buf_elems);
    const int P = 32;   // one workgroup per CU of one XCD
    float best[3] = {1e30f, 1e30f, 1e30f};
    int failed = 0;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipMemset(sync, 0, sizeof(ChainSync)));
            for (int b = 1; b <= n_ops; ++b) CK(hipMemset(bufs[b], 0, buf_elems * 2));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (mode == 0)
                for (int p = 0; p < n_ops; ++p) one_kernel<WM, WN, WK, MREP, NREP, LW><<<grid, v.threads, lds>>>(dprobs, p);
            else if (mode == 1)
                chain_kernel<WM, WN, WK, MREP, NREP, LW><<<grid, v.threads, lds>>>(dprobs, n_ops, sync);
            else
                chain_xcd_kernel<WM, WN, WK, MREP, NREP, LW><<<8 * P, v.threads, lds>>>(dprobs, n_ops, sync, P);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (rep > 1 && t < best[mode]) best[mode] = t;
        }
        CK(hipMemcpy(mode == 0 ? out_a.data() : mode == 1 ? out_b.data() : out_c.data(), bufs[n_ops], buf_elems * 2, hipMemcpyDeviceToHost));
        ChainSync h;
        CK(hipMemcpy(&h, sync, sizeof(h), hipMemcpyDeviceToHost));
        failed |= (int)h.fail;
    }
    const bool same = std::memcmp(out_a.data(), out_b.data(), buf_elems * 2) == 0;
    double sum = 0;
    for (size_t i = 0; i < buf_elems; ++i) sum += std::fabs((double)__half2float(out_a[i]));
    const bool same_c = std::memcmp(out_a.data(), out_c.data(), buf_elems * 2) == 0;
    std::printf("%-28s grid %3d x %3d threads: three launches %6.2f us   one launch with two grid barriers %6.2f us   (%+5.1f %%)   outputs %s%s  [mean |y| %.4f]\n", label,
                grid, v.threads, best[0] * 1e3, best[1] * 1e3, (best[1] / best[0] - 1.0) * 100.0, same ? "bit-identical" : "DIFFER", failed ? "  BARRIER TIMED OUT" : "",
                sum / buf_elems);
    std::printf("%-28s   ... on ONE XCD (32 workgroups walking %d tiles per layer, L2-local barrier): %6.2f us   (%+5.1f %% against the three launches)   outputs %s\n", "",
                grid, best[2] * 1e3, (best[2] / best[0] - 1.0) * 100.0, same_c ? "bit-identical" : "DIFFER");
    CK(hipFree(dprobs));
    CK(hipFree(sync));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int chain_main() {
    const int n = 1, h = 20, w = 20, c = 288, reps = 30;
    const size_t elems = (size_t)n * h * w * c;
    __half* bufs[4];
    for (auto& b : bufs) CK(hipMalloc(&b, elems * 2));
    unsigned seed = 12345u;
    const auto rnd = [&] {
        seed = seed * 1664525u + 1013904223u;
        return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f;
    };
    std::vector<__half> x(elems);
    for (auto& v : x) v = __float2half(rnd());
    CK(hipMemcpy(bufs[0], x.data(), elems * 2, hipMemcpyHostToDevice));
    std::vector<Layer> layers(3);
    for (int p = 0; p < 3; ++p) {
        std::vector<float> wf((size_t)c * c * 9), b(c);
        const float ws = 2.0f / std::sqrt((float)c * 9);
        for (auto& v : wf) v = rnd() * ws;
        for (auto& v : b) v = rnd() * 0.5f;
        ConvArgs a{};
        std::vector<__half> hw, p32;
        pack_conv_weights(wf.data(), c, c, 3, 3, c, c, hw, a.K, a.Kp);
        pack_conv_weights_t32(hw.data(), c, c, a.Kp, p32, 9);
        Layer& L = layers[p];
        CK(hipMalloc(&L.w, hw.size() * 2));
        CK(hipMalloc(&L.w32, p32.size() * 2));
        CK(hipMalloc(&L.bias, b.size() * 4));
        CK(hipMemcpy(L.w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(L.w32, p32.data(), p32.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(L.bias, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        a.in = bufs[p];
        a.in_cs = c, a.in_co = 0;
        a.N = n, a.H = h, a.W = w, a.Cin = c, a.Ho = h, a.Wo = w, a.KH = 3, a.KW = 3, a.stride = 1, a.pad = 1;
        a.wt = L.w, a.bias = L.bias;
        a.out = bufs[p + 1], a.out_cs = c, a.out_co = 0;
        if (p == 1) a.res = bufs[0], a.res_cs = c, a.res_co = 0;   // the bottleneck's shortcut
        a.Cout_pad = c, a.M = n * h * w, a.act = 1;
        a.in_bytes = (unsigned)(elems * 2), a.wt_bytes = (unsigned)(hw.size() * 2);
        a.wt_t32 = L.w32, a.wt_t32_bytes = (unsigned)(p32.size() * 2);
        L.a = a;
    }
    std::printf("three dependent 3x3 layers, one image, 20 x 20 x 288 (M 400, N 288, K 2592 each; the second adds the first's input): conv_sb's tile code (sb_tile)\n");
    run_shape<1, 1, 1, 1, 1, 0>("32 x 32 tiles, one wave", layers, bufs, elems, reps);
    run_shape<2, 1, 1, 1, 1, 0>("64 x 32 tiles, two waves", layers, bufs, elems, reps);
    run_shape<4, 1, 1, 1, 1, 0>("128 x 32 tiles, four waves", layers, bufs, elems, reps);
    run_shape<4, 1, 1, 1, 3, 0>("128 x 96 tiles, four waves", layers, bufs, elems, reps);
    // the product's own forms where every wave of a workgroup passes the same number of barriers inside a tile (K shared by
    // three waves without loaders: variants 0 and 2; loaders without K sharing: variant 52)
    run_shape<1, 1, 3, 1, 1, 0>("32 x 32, K over three waves", layers, bufs, elems, reps);
    run_shape<2, 1, 3, 1, 1, 0>("64 x 32, K over three waves", layers, bufs, elems, reps);
    run_shape<4, 1, 1, 1, 3, 4>("128 x 96, 4 loaders + 4", layers, bufs, elems, reps);
    // the product's launches of the same three layers (its tuner's variants for this layer: loaders, K shared by three waves)
    DeviceCtx& ctx = device_ctx(0);
    for (int variant : {44, 60, 40, 0}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        float best = 1e30f;
        if (!conv_sb_supported(layers[0].a, variant)) continue;
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, ctx.stream));
            for (int p = 0; p < 3; ++p) launch_conv_sb(ctx, ctx.stream, layers[p].a, variant);
            CK(hipEventRecord(e1, ctx.stream));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (rep > 1 && t < best) best = t;
        }
        std::printf("product variant %2d (%3d x %2d tiles, %d threads): three launches %6.2f us\n", variant, conv_sb_tile(variant).bm, conv_sb_tile(variant).bn,
                    kSbVariants[variant].threads, best * 1e3);
    }
    return 0;
}

}  // namespace
}  // namespace rmr

int main() { return rmr::chain_main(); }
