// conv_ws.hip -- weights-stationary 3x3 / stride 1 / pad 1 convolution for narrow layers
// (Cin = Cout = 48 on 160-wide maps: the P2-level C2f bottlenecks of YOLOv8m).
//
// With only 48 output channels an implicit-GEMM tile reuses each staged input pixel 48 times, so the
// tiled kernels (conv_igemm / conv_dma / conv_halo) spend their time re-streaming the 41 KiB of
// weights through LDS once per 128..384 pixels and synchronising every 32-deep K slice: 220-240
// TFLOP/s measured.  Here the whole filter lives in REGISTERS instead:
//
//   * one 4-wave workgroup per CU (1 wave per SIMD, up to 512 VGPRs each); every wave loads all
//     48 x 432 weights as MFMA B fragments once (14 K steps x 3 channel tiles = 168 VGPRs) and
//     keeps them for its whole strip of the image;
//   * the workgroup walks a strip of image rows top to bottom, two output rows per step.  Input
//     rows are DMA'd (buffer_load ... lds) into an 8-slot LDS ring of full rows, each row fetched
//     ONCE per strip (plus one halo row at each end), two steps ahead of its use; one barrier per
//     step (210 MFMAs per wave) instead of one per 12-24 MFMAs;
//   * a ring row is [zero pixel][160 pixels][zero pixel] of 96 bytes each, so the kw = 0 / 2 taps
//     at the image edge read zeros and rows above / below the image arrive as zeros from the
//     buffer bounds check: no validity masks.  The 96-byte pixel pitch is bank-conflict-free for
//     ds_read_b128 fragment reads as it stands (24 p mod 64 visits 8 distinct octets, and the two
//     k-groups of a lane group sit in different halves of an octet);
//   * K runs over (tap, channel) = 432 = 13.5 MFMA K steps: a K step may straddle two taps, which
//     only means that lanes 0-31 and 32-63 read at different (row, column) shifts.
//
// Each wave owns half a row (80 pixels = 5 MFMA column tiles) x all 48 channels: per K step
// 5 fragment reads feed 15 MFMAs, and nothing but activations moves after the prologue.
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

constexpr int WS_C = 48;                       // input = output channels
constexpr int WS_W = 160;                      // map width
constexpr int WS_PIX = WS_C * 2;               // bytes per pixel in LDS
constexpr int WS_ROW = (WS_W + 2) * WS_PIX;    // ring slot: zero pixel, row, zero pixel
constexpr int WS_SLOTS = 8;
constexpr int WS_KSTEPS = 14;                  // ceil(9 * 48 / 32)
constexpr int WS_DMA_ROW = WS_W * WS_PIX / 1024;  // 15 DMA instructions per row
// NJ = 1: 4 waves, each all 48 output channels (weights 168 VGPRs, one wave per SIMD).
// NJ = 3: 12 waves, wave (row, half, j) computes channel tile j only (weights 56 VGPRs, three waves
// per SIMD): every input fragment is read three times from LDS, but one wave's epilogue and waits
// now hide under the other two waves' MFMAs.
constexpr int ws_ni(int nj) { return (2 * WS_DMA_ROW + 4 * nj - 1) / (4 * nj); }  // DMA instructions per wave per row pair
constexpr int ws_stage(int nj) { return ws_ni(nj) * 1024; }  // per-wave stage: 80 pixels x (96 / NJ) bytes, padded to whole DMA KiB
constexpr int ws_lds(int nj) { return WS_SLOTS * WS_ROW + 4 * nj * ws_stage(nj) + 1024; }  // + one KiB that idle DMA slots land in

// v * rcp(1 + e^-v): the hardware reciprocal (1 ulp) instead of an IEEE division, 60 per step per lane
__device__ __forceinline__ float silu_w(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ void dma16w(u32x4 rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmw() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// strip_rows: output rows per workgroup (even, divides H)
template <bool ACT, bool RES, bool OUT32, int NJ>
__global__ __launch_bounds__(256 * NJ) __attribute__((amdgpu_waves_per_eu(NJ, NJ)))
void conv_ws_kernel(const ConvArgs a, const int strip_rows) {
    constexpr int NW = 4 * NJ;              // waves
    constexpr int NT = 3 / NJ;              // 16-channel tiles per wave
    constexpr int WS_NI = ws_ni(NJ);
    constexpr int WS_STAGE = ws_stage(NJ);
    constexpr int CPP = NT * 2;             // 16-byte chunks per pixel in a wave's stage
    static_assert(NJ == 1 || NJ == 3, "1 or 3 wave groups along the output channels");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / NJ, jw = wave % NJ;
    const int r = grp >> 1;           // output row of the step this wave computes
    const int xh = grp & 1;           // left / right half of the row
    const int frow = lane & 15;
    const int kg = lane >> 4;
    const bool hi = kg >= 2;          // lanes 32-63 take the second 16 k of a K step

    const int strips = a.H / strip_rows;
    const int img = blockIdx.x / strips;
    const int y_base = (blockIdx.x % strips) * strip_rows;
    const int steps = strip_rows / 2;

    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu),
                           sgpr(a.in_bytes), sgpr(0x00020000u)};
    const unsigned scratch = sgpr(lds0 + WS_SLOTS * WS_ROW + NW * WS_STAGE);

    // ---- zero the edge pixels of every ring slot (never written again) -------------------------
    for (int i = tid; i < WS_SLOTS * 2 * (WS_PIX / 16); i += 64 * NW) {
        const int slot = i / (2 * (WS_PIX / 16));
        const int rem = i % (2 * (WS_PIX / 16));
        const int side = rem / (WS_PIX / 16), c16 = rem % (WS_PIX / 16);
        *(u32x4*)(smem + slot * WS_ROW + side * (WS_W + 1) * WS_PIX + c16 * 16) = u32x4{0, 0, 0, 0};
    }
    __syncthreads();

    // ---- the filter, as B fragments, for the whole kernel ---------------------------------------
    half8 wreg[WS_KSTEPS][NT];
#pragma unroll
    for (int ks = 0; ks < WS_KSTEPS; ++ks)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            wreg[ks][j] = *(const half8*)((const _Float16*)a.wt + (size_t)((jw * NT + j) * 16 + frow) * a.Kp + ks * 32 + kg * 8);
    // opaque to the optimiser from here on: otherwise it re-loads fragments from memory inside the
    // step loop (rematerialisation) instead of keeping them in registers
#pragma unroll
    for (int ks = 0; ks < WS_KSTEPS; ++ks)
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(wreg[ks][j]));

    // ---- DMA bookkeeping: relative row ry (0 = y_base - 1) lives in ring slot ry % 8 ----------
    // slot q = wave + NW j of a row pair: row q / 15 of the pair, instruction q % 15 of that row
    unsigned goff[WS_NI];
#pragma unroll
    for (int j = 0; j < WS_NI; ++j) {
        const int q = wave + NW * j;
        const int g = (q % WS_DMA_ROW) * 64 + lane;  // 16-byte chunk within the row
        goff[j] = (unsigned)(((g / 6) * a.in_cs + a.in_co + (g % 6) * 8) * 2);
    }
    const int img_row0 = img * a.H;
    // wave-uniform constants of each slot, then a branch-free issue (selects as mask arithmetic:
    // the optimiser otherwise turns the three conditions into a chain of scalar branches)
    unsigned q_ok[WS_NI], q_row[WS_NI], q_dst[WS_NI];
#pragma unroll
    for (int j = 0; j < WS_NI; ++j) {
        const int q = wave + NW * j;
        q_ok[j] = q < 2 * WS_DMA_ROW ? 0xffffffffu : 0u;
        q_row[j] = q >= WS_DMA_ROW ? 0xffffffffu : 0u;
        q_dst[j] = (unsigned)(WS_PIX + (q % WS_DMA_ROW) * 1024);
    }
    // a row pair's wave-uniform constants, then one DMA per slot j: the step loop issues the slots one by one behind
    // its MFMAs, the prologue all at once
    struct Pair {
        unsigned live0, live1, rowoff0, slot0, slot1;
    };
    const unsigned row_bytes = (unsigned)(WS_W * a.in_cs * 2);
    auto pair_of = [&](int ry0) {  // rows ry0, ry0 + 1
        const int gy0 = y_base - 1 + ry0;
        Pair p;
        p.live0 = (gy0 >= 0 && gy0 < a.H && ry0 <= strip_rows + 1) ? 0xffffffffu : 0u;
        p.live1 = (gy0 + 1 >= 0 && gy0 + 1 < a.H && ry0 + 1 <= strip_rows + 1) ? 0xffffffffu : 0u;
        p.rowoff0 = (unsigned)(img_row0 + gy0) * row_bytes;
        p.slot0 = lds0 + (unsigned)(ry0 % WS_SLOTS) * WS_ROW;
        p.slot1 = lds0 + (unsigned)((ry0 + 1) % WS_SLOTS) * WS_ROW;
        return p;
    };
    auto issue_slot = [&](const Pair& p, int j) {  // j: compile-time at every call site
        const unsigned live = q_ok[j] & ((q_row[j] & p.live1) | (~q_row[j] & p.live0));
        const unsigned rowoff = p.rowoff0 + (q_row[j] & row_bytes);
        const unsigned off = (goff[j] + rowoff) | ~live;  // dead slots: offset 0xffffffff is out of range
        const unsigned slot = (q_row[j] & p.slot1) | (~q_row[j] & p.slot0);
        const unsigned dst = (q_ok[j] & (slot + q_dst[j])) | (~q_ok[j] & scratch);
        dma16w(in_rsrc, sgpr(dst), off);
    };
    auto issue_pair = [&](int ry0) {
        const Pair p = pair_of(ry0);
#pragma unroll
        for (int j = 0; j < WS_NI; ++j) issue_slot(p, j);
    };
    issue_pair(0);
    issue_pair(2);
    issue_pair(4);

    const unsigned lane_off = (unsigned)(frow * WS_PIX + (kg & 1) * 16 + xh * 80 * WS_PIX);
    const int px = lane & 15;
    const int cq = (lane >> 4) * 4;

    // ---- output / residual staging: this wave's 80 pixels x 96 B, pixel-major ------------------
    // MFMA results come out as 8-byte pieces (4 channels of one pixel per lane): stored like that
    // they are 32-byte fragments of cache lines.  They go through LDS instead and leave as 16-byte
    // chunks, six consecutive lanes per pixel; the residual comes IN the same way (LDS-DMA into
    // the stage, added in place).  A wave only ever touches its own stage, so no barrier is needed.
    const unsigned stage = sgpr(lds0 + WS_SLOTS * WS_ROW + wave * WS_STAGE);
    unsigned char* const stage_p = smem + WS_SLOTS * WS_ROW + wave * WS_STAGE;
    unsigned ooff[WS_NI], roff[WS_NI], cmask[WS_NI];  // per chunk c = lane + 64 t of the stage
#pragma unroll
    for (int t = 0; t < WS_NI; ++t) {
        const int c = lane + 64 * t;
        cmask[t] = c < 80 * CPP ? 0xffffffffu : 0u;
        ooff[t] = (unsigned)(((c / CPP) * a.out_cs + a.out_co + jw * NT * 16 + (c % CPP) * 8) * 2);
        roff[t] = (unsigned)(((c / CPP) * a.res_cs + a.res_co + jw * NT * 16 + (c % CPP) * 8) * 2);
    }
    const u32x4 res_rsrc = {sgpr((unsigned)(size_t)a.res), sgpr((unsigned)((size_t)a.res >> 32) & 0xffffu),
                            sgpr(0xffffffffu), sgpr(0x00020000u)};
    long m_row = 0;
    // the previous step's results leave as 16-byte pieces: read from the stage at the step's start (before the
    // residual DMAs overwrite it), stored one by one behind the MFMAs through a bounds-checked resource -- a piece
    // without a pixel, or of the step "before the first", goes to an out-of-range offset, so the count is constant
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0xfffffff0u, 0x00020000);
    u32x4 dv[WS_NI];
    unsigned out_row = 0xffffffffu;   // byte offset of the previous step's first pixel; all ones: there is none
    auto drain_reads = [&]() {
        if (OUT32) return;
#pragma unroll
        for (int t = 0; t < WS_NI; ++t) dv[t] = *(const u32x4*)(stage_p + (lane + 64 * t) * 16);
    };
    auto drain_store = [&](int t) {  // t: compile-time at every call site
        if (OUT32) return;
        const unsigned off = (out_row == 0xffffffffu ? 0xffffffffu : out_row + ooff[t]) | ~cmask[t];
        __builtin_amdgcn_raw_buffer_store_b128(dv[t], out_rsrc, off, 0, 0);
    };

    float4 bias[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bias[j] = *(const float4*)(a.bias + (jw * NT + j) * 16 + cq);

    // fragment reads of K step ks: k = 32 ks + 8 kg, lanes 0-31 start at kL, lanes 32-63 at
    // kH = kL + 16 (same or next tap); k >= 432 meets zero weights, so it re-reads finite data
    unsigned vb[3];
    auto read_frags = [&](int ks, half8* xf) {
        const int kL = 32 * ks, kH = 32 * ks + 16;
        const int tapL = kL / WS_C, cL = kL % WS_C;
        const int tapH = kH < 9 * WS_C ? kH / WS_C : tapL, cH = kH < 9 * WS_C ? kH % WS_C : cL;
        const unsigned immL = (unsigned)((tapL % 3) * WS_PIX + cL * 2);
        const unsigned immH = (unsigned)((tapH % 3) * WS_PIX + cH * 2);
        const unsigned addr = hi ? vb[tapH / 3] + immH : vb[tapL / 3] + immL;
        const __attribute__((address_space(3))) unsigned char* p =
            (const __attribute__((address_space(3))) unsigned char*)(size_t)addr;
#pragma unroll
        for (int i = 0; i < 5; ++i) xf[i] = *(const __attribute__((address_space(3))) half8*)(p + i * 16 * WS_PIX);
    };

    // Vector-memory instructions of a step, in issue order: the residual DMAs (needed by this step's epilogue), the
    // row-pair DMAs (needed two steps on), the stores of the previous step's results.  They are NOT issued in a block
    // at the step's start (a vector-memory instruction costs its wave 60-180 issue cycles and the MFMA pipe runs dry
    // meanwhile) but NVM_PER at a time behind the MFMAs of the K steps; the waits count what may still be in flight
    // (loads and stores share vmcnt and retire in order).
    constexpr int N_RES = RES ? WS_NI : 0;
    constexpr int N_ST = OUT32 ? 0 : WS_NI;
    constexpr int NVM = N_RES + WS_NI + N_ST;
    constexpr int NVM_PER = (NVM + WS_KSTEPS - 1) / WS_KSTEPS;
    static_assert(NVM <= 63, "vmcnt is 6 bits");
    for (int s = 0; s < steps; ++s) {
        // The rows of this step were issued two steps ago (or by the prologue): everything older than the previous
        // step's own instructions has landed.  (Step 0: the prologue's third pair may still be in flight.)
        if (s == 0)
            wait_vmw<WS_NI>();
        else
            wait_vmw<NVM>();
        __builtin_amdgcn_s_barrier();    // also: step s - 1 is fully consumed by every wave
        drain_reads();
        out_row = s > 0 ? (unsigned)(m_row * a.out_cs * 2) : 0xffffffffu;
        const Pair pair = pair_of(2 * s + 6);   // rows of step s + 2 (slots last read in step s - 1)
        const int y = y_base + 2 * s + r;
        m_row = ((long)img_row0 + y) * WS_W + xh * 80;
        const unsigned rbase = RES ? (unsigned)(m_row * a.res_cs * 2) : 0u;
        if (RES) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage is read before the residual lands in it
        const auto vm_op = [&](int v) {  // v: compile-time at every call site
            if (v < N_RES)  // this wave's residual pixels -> its stage
                dma16w(res_rsrc, sgpr(stage + v * 1024), (roff[v < N_RES ? v : 0] + rbase) | ~cmask[v < N_RES ? v : 0]);
            else if (v < N_RES + WS_NI)
                issue_slot(pair, v - N_RES);
            else if (v < NVM)
                drain_store(v - N_RES - WS_NI);
        };

        // LDS address of this lane's pixel column in the three input rows of its output row
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) vb[kh] = lds0 + ((2 * s + r + kh) % WS_SLOTS) * WS_ROW + lane_off;

        floatx4 acc[5][NT];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

        half8 xf[2][5];
        read_frags(0, xf[0]);
#pragma unroll
        for (int ks = 0; ks < WS_KSTEPS; ++ks) {
            if (ks + 1 < WS_KSTEPS) read_frags(ks + 1, xf[(ks + 1) & 1]);  // next K step's reads ride under these MFMAs
            __builtin_amdgcn_sched_barrier(0);
            constexpr int HALF = (5 * NT + 1) / 2;
#pragma unroll
            for (int c = 0; c < HALF; ++c)
                acc[c / NT][c % NT] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[ks][c % NT], xf[ks & 1][c / NT], acc[c / NT][c % NT], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = ks * NVM_PER; v < (ks + 1) * NVM_PER; ++v) vm_op(v);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = HALF; c < 5 * NT; ++c)
                acc[c / NT][c % NT] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[ks][c % NT], xf[ks & 1][c / NT], acc[c / NT][c % NT], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue of the step: bias, SiLU, residual -> packed f16 into the stage --------------
        if (RES) wait_vmw<WS_NI + N_ST>();  // own residual DMAs (and everything older) have landed; the pair DMAs and stores behind them need not
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = (jw * NT + j) * 16 + cq;
                const float4 b = bias[j];
                float v[4] = {acc[i][j][0] + b.x, acc[i][j][1] + b.y, acc[i][j][2] + b.z, acc[i][j][3] + b.w};
                if (ACT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_w(v[e]);
                }
                unsigned char* const sp = stage_p + (i * 16 + px) * (CPP * 16) + (j * 16 + cq) * 2;
                if (RES) {
                    union {
                        uint2 u;
                        _Float16 h[4];
                    } rr;
                    rr.u = *(const uint2*)sp;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)rr.h[e];
                }
                if (OUT32) {  // f32 view (the kernel's parity test): stored at once
                    *(float4*)(a.out32 + (m_row + i * 16 + px) * a.out_cs + a.out_co + n) = make_float4(v[0], v[1], v[2], v[3]);
                    continue;
                }
                union {
                    uint2 u;
                    _Float16 h[4];
                } o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o.h[e] = (_Float16)v[e];
                *(uint2*)sp = o.u;
            }
        }
    }
    // the strip's last step
    drain_reads();
    out_row = (unsigned)(m_row * a.out_cs * 2);
#pragma unroll
    for (int t = 0; t < WS_NI; ++t) drain_store(t);
    wait_vmw<0>();
}

// variant = strip height x wave layout (NJ = 1 for ids 0..5, NJ = 3 for ids 6..11)
const int kWsStripRows[] = {40, 20, 10, 8, 4, 2};
constexpr int kNumStrips = sizeof(kWsStripRows) / sizeof(kWsStripRows[0]);
constexpr int kNumWs = 2 * kNumStrips;

}  // namespace

int conv_ws_num_variants() { return kNumWs; }

bool conv_ws_supported(const ConvArgs& a, int variant) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1) return false;
    if (a.Cin != WS_C || a.Cout_pad != WS_C || a.W != WS_W || a.Wo != a.W || a.Ho != a.H) return false;
    if (a.Kp < WS_KSTEPS * 32 || (!a.out32 && !a.out)) return false;
    if (variant < 0) return a.H % 2 == 0;
    return variant < kNumWs && a.H % kWsStripRows[variant % kNumStrips] == 0;
}

void launch_conv_ws(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int variant) {
    if (variant < 0 || variant >= kNumWs) fail(RMR_ERR_INVALID_ARGUMENT, "conv_ws: variant %d out of range", variant);
    if (!conv_ws_supported(a, variant)) fail(RMR_ERR_LOGIC, "conv_ws: layer not supported by variant %d", variant);
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 8 || a.out_co % 8 || a.res_cs % 8 || a.res_co % 8)
        fail(RMR_ERR_LOGIC, "conv_ws: misaligned view");
    if (a.in_bytes == 0 || a.in_bytes > 0xf0000000ull) fail(RMR_ERR_LOGIC, "conv_ws: input view size not set or larger than 3.75 GiB");
    if (!a.out32 && (double)a.M * a.out_cs * 2 >= 4.0e9) fail(RMR_ERR_LOGIC, "conv_ws: output view of 4 GB or more (32-bit store offsets)");
    using Kern = void (*)(const ConvArgs, int);
#define WS_KERNELS(NJ)                                                                                   \
    {conv_ws_kernel<false, false, false, NJ>, conv_ws_kernel<false, false, true, NJ>,                    \
     conv_ws_kernel<false, true, false, NJ>,  conv_ws_kernel<false, true, true, NJ>,                     \
     conv_ws_kernel<true, false, false, NJ>,  conv_ws_kernel<true, false, true, NJ>,                     \
     conv_ws_kernel<true, true, false, NJ>,   conv_ws_kernel<true, true, true, NJ>}
    static const Kern kernels[2][8] = {WS_KERNELS(1), WS_KERNELS(3)};
#undef WS_KERNELS
    static std::once_flag once;
    std::call_once(once, [] {
        for (const auto& row : kernels)
            for (Kern k : row)
                (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int nj = variant < kNumStrips ? 1 : 3;
    const Kern kernel = kernels[nj == 1 ? 0 : 1][(a.act ? 4 : 0) + (a.res ? 2 : 0) + (a.out32 ? 1 : 0)];
    const int sr = kWsStripRows[variant % kNumStrips];
    const int grid = a.N * (a.H / sr);
    const double flops = a.flops > 0 ? a.flops : 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad + (double)a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d w%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, variant);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    kernel<<<grid, 256 * nj, ws_lds(nj), stream>>>(a, sr);
    RMR_HIP(hipGetLastError());
}

}  // namespace rmr
