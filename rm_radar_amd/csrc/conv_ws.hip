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
#include <type_traits>

#include "conv_igemm.h"

namespace rmr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));

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

// =====================================================================================================================
// conv_wsp_kernel (round 4): the same weights-stationary walk, software-pipelined per 16-pixel tile.
//
// What the kernel above loses (measured, 256 images, w0: 456 us per launch = 13 000 cycles per two-row step against 3 360
// cycles of MFMA work per SIMD; without the SiLU 365 us; with a shortcut 541 us): its wave runs K-outer -- 14 K steps over
// five pixel tiles -- so the 60 epilogue values per lane (two transcendentals each), the 24 vector-memory instructions of a
// step (60-180 issue cycles each) and the LDS traffic of the stage all sit in the SAME in-order instruction stream as the
// MFMAs: with one wave per SIMD nothing runs beside anything.  Here
//   * a wave runs TILE-outer: the 42 MFMAs of tile g (14 K steps x 3 channel tiles) carry, as fillers between them, the bias + SiLU of tile g - 1 (one value per K step, results into the wave's
//     own 3 KiB stage) and the drain of tile g - 2 (stage -> 16-byte chunks, + the shortcut pixels fetched as
//     16-byte chunks into registers one tile earlier, -> two bounds-checked stores): every stage of a tile's life rides
//     under another tile's MFMAs, and the f32 sum SiLU + shortcut is still rounded to f16 once (f32 stage);
//   * only 12 accumulators are live at a time, so a wave needs ~250 registers with the whole filter resident (168)
//     and TWO waves fit a SIMD (NSPLIT = 2: waves w and w + 4 share the half row of SIMD w, three tiles + two tiles):
//     one wave's vector-memory issue and transcendentals run beside the other's MFMAs;
//   * same ring of rows, same DMAs, same barrier per step; the same f32 operation order per output value as the kernel
//     above (K steps in order on one accumulator, then the bias), so the two are bit-identical.
// vmcnt discipline (loads and stores retire in order): the shortcut loads of a tile are issued after that tile loop's two
// stores and before its row DMAs, and are waited for one tile loop later with vmcnt(row DMAs issued since).
constexpr int WP_SLOTS = 8;
// The stage's pixel pitch is NOT the pixel's size: at 96 bytes (f16) pixels p and p + 8 start on the same bank, at 192 bytes
// (f32) pixels p and p + 4 -- the sixteen pixels of a stage write met on two / four bank groups, 45 % of the kernel's LDS cycles
// were bank conflicts (profiles/r04_conv_ws_pmc.txt).  112 = 7 x 16 and 208 = 13 x 16 bytes put the sixteen pixels of a
// ds_write_b64 / ds_write_b128 pass on sixteen different bank groups and keep the drain's 16-byte reads aligned.
constexpr int WP_PITCH16 = 112, WP_PITCH32 = 208;
constexpr int WP_STAGE = 16 * WP_PITCH32;   // per wave: one tile of 16 pixels x 48 channels as f32 (shortcut) or f16
constexpr int wp_lds(int nw) { return WP_SLOTS * WS_ROW + nw * WP_STAGE + 1024 + 256; }  // + idle-DMA KiB + bias

__device__ __forceinline__ u32x4 ld16_asm(u32x4 rsrc, unsigned voff) {
    u32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
    return v;
}

// ABL (development builds, -DRMR_WSP_ABLATE + RMR_WSP_ABLATE=<mask> in the environment): parts removed to price them --
// 1 no MFMAs, 2 no stores, 4 row DMAs out of range (issued, nothing fetched), 8 no fragment reads, 16 no transcendentals,
// 32 shortcut loads out of range, 64 no barrier
template <bool ACT, bool RES, bool OUT32, int NSPLIT, int ABL = 0>
__global__ __launch_bounds__(256 * NSPLIT) void conv_wsp_kernel(const ConvArgs a, const int strip_rows) {
    constexpr int NW = 4 * NSPLIT;
    constexpr int NI = (2 * WS_DMA_ROW + NW - 1) / NW;   // row-pair DMA instructions per wave and step: 8 / 4
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave & 3;          // the half row of SIMD `grp` (waves w and w + 4 share a SIMD)
    const int sub = wave >> 2;         // NSPLIT = 2: 0 = tiles 0..2 of the half row, 1 = tiles 3..4
    const int r = grp >> 1, xh = grp & 1;
    const int t0 = NSPLIT == 2 && sub ? 3 : 0;
    const int frow = lane & 15, kg = lane >> 4;
    const bool hi = kg >= 2;

    const int strips = a.H / strip_rows;
    const int img = blockIdx.x / strips;
    const int y_base = (blockIdx.x % strips) * strip_rows;
    const int steps = strip_rows / 2;

    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu),
                           sgpr((ABL & 4) ? 0u : a.in_bytes), sgpr(0x00020000u)};
    unsigned char* const stage_p = smem + WP_SLOTS * WS_ROW + wave * WP_STAGE;
    const unsigned scratch = sgpr(lds0 + WP_SLOTS * WS_ROW + NW * WP_STAGE);
    float* const bias_p = (float*)(smem + WP_SLOTS * WS_ROW + NW * WP_STAGE + 1024);

    for (int i = tid; i < WP_SLOTS * 2 * (WS_PIX / 16); i += 64 * NW) {
        const int slot = i / (2 * (WS_PIX / 16));
        const int rem = i % (2 * (WS_PIX / 16));
        const int side = rem / (WS_PIX / 16), c16 = rem % (WS_PIX / 16);
        *(u32x4*)(smem + slot * WS_ROW + side * (WS_W + 1) * WS_PIX + c16 * 16) = u32x4{0, 0, 0, 0};
    }
    if (tid < WS_C) bias_p[tid] = a.bias[tid];
    __syncthreads();

    half8 wreg[WS_KSTEPS][3];
#pragma unroll
    for (int ks = 0; ks < WS_KSTEPS; ++ks)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            wreg[ks][j] = *(const half8*)((const _Float16*)a.wt + (size_t)(j * 16 + frow) * a.Kp + ks * 32 + kg * 8);
#pragma unroll
    for (int ks = 0; ks < WS_KSTEPS; ++ks)
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(wreg[ks][j]));

    // ---- row-pair DMAs: as in the kernel above, NW waves share the 30 instructions of a pair ----
    unsigned goff[NI], q_ok[NI], q_row[NI], q_dst[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int q = wave + NW * j;
        const int g = (q % WS_DMA_ROW) * 64 + lane;
        goff[j] = (unsigned)(((g / 6) * a.in_cs + a.in_co + (g % 6) * 8) * 2);
        q_ok[j] = q < 2 * WS_DMA_ROW ? 0xffffffffu : 0u;
        q_row[j] = q >= WS_DMA_ROW ? 0xffffffffu : 0u;
        q_dst[j] = (unsigned)(WS_PIX + (q % WS_DMA_ROW) * 1024);
    }
    const int img_row0 = img * a.H;
    struct Pair {
        unsigned live0, live1, rowoff0, slot0, slot1;
    };
    const unsigned row_bytes = (unsigned)(WS_W * a.in_cs * 2);
    auto pair_of = [&](int ry0) {
        const int gy0 = y_base - 1 + ry0;
        Pair p;
        p.live0 = (gy0 >= 0 && gy0 < a.H && ry0 <= strip_rows + 1) ? 0xffffffffu : 0u;
        p.live1 = (gy0 + 1 >= 0 && gy0 + 1 < a.H && ry0 + 1 <= strip_rows + 1) ? 0xffffffffu : 0u;
        p.rowoff0 = (unsigned)(img_row0 + gy0) * row_bytes;
        p.slot0 = lds0 + (unsigned)(ry0 % WP_SLOTS) * WS_ROW;
        p.slot1 = lds0 + (unsigned)((ry0 + 1) % WP_SLOTS) * WS_ROW;
        return p;
    };
    auto issue_slot = [&](const Pair& p, int j) {
        const unsigned live = q_ok[j] & ((q_row[j] & p.live1) | (~q_row[j] & p.live0));
        const unsigned rowoff = p.rowoff0 + (q_row[j] & row_bytes);
        const unsigned off = (goff[j] + rowoff) | ~live;
        const unsigned slot = (q_row[j] & p.slot1) | (~q_row[j] & p.slot0);
        const unsigned dst = (q_ok[j] & (slot + q_dst[j])) | (~q_ok[j] & scratch);
        dma16w(in_rsrc, sgpr(dst), off);
    };
    auto issue_pair = [&](int ry0) {
        const Pair p = pair_of(ry0);
#pragma unroll
        for (int j = 0; j < NI; ++j) issue_slot(p, j);
    };
    issue_pair(0);
    issue_pair(2);
    issue_pair(4);

    // ---- per-lane constants ----
    // fragment reads: within one filter row the im2col index k' = kw * 48 + ci is LINEAR in the ring row's bytes (taps are
    // neighbouring 96-byte pixels), so a lane's address is base(filter row) + 2 k' and lane group kg adds 16 kg; only K
    // step 4 (k = 128..159: lanes 0-31 in filter row 0, lanes 32-63 in row 1) and step 13 (k >= 432: zero weights, lanes
    // 32-63 re-read what lanes 0-31 read) select per lane.
    const unsigned lane_off = (unsigned)(frow * WS_PIX + kg * 16 + (xh * 80 + t0 * 16) * WS_PIX);
    const int px = lane & 15, cq = (lane >> 4) * 4;
    // MFMA layout -> stage (pixel-major; f32 when a shortcut is added at the drain, else f16)
    unsigned char* const sw = stage_p + (RES ? px * WP_PITCH32 + cq * 4 : px * WP_PITCH16 + cq * 2);
    // drain layout: chunk c = 16 bytes of output = 8 channels; six chunks per pixel; lane l drains chunks l and 64 + l (l < 32)
    // (pixel, part) of the two chunks packed into ONE register: the byte offsets into the stage, the output and the
    // shortcut tensor are two multiply-adds away when they are needed (registers are what limits two waves per SIMD)
    const int c0 = lane, c1 = (64 + lane) % 96;
    const unsigned pq = (unsigned)((c0 / 6) | ((c0 % 6) << 4) | ((c1 / 6) << 8) | ((c1 % 6) << 12) | (lane >= 32 ? 1 << 16 : 0));
    // (read through an empty asm at every use: otherwise the optimiser hoists the derived offsets out of the step loop and
    // they occupy the registers this packing is meant to free)
    const auto pqv = [&]() {
        unsigned q = pq;
        asm volatile("" : "+v"(q));
        return q;
    };
    const auto cp = [&](int c) { return (pqv() >> (c ? 8 : 0)) & 15u; };
    const auto cqq = [&](int c) { return (pqv() >> (c ? 12 : 4)) & 15u; };
    const auto dead = [&](int c) { return c ? 0u - ((pqv() >> 16) & 1u) : 0u; };   // all ones: lanes 32-63 have no second chunk
    const auto stage_rd = [&](int c) { return stage_p + (RES ? cp(c) * (unsigned)WP_PITCH32 + cqq(c) * 32u : cp(c) * (unsigned)WP_PITCH16 + cqq(c) * 16u); };
    const unsigned out_pitch2 = (unsigned)a.out_cs * 2u, out_co2 = (unsigned)a.out_co * 2u;
    const unsigned res_pitch2 = (unsigned)a.res_cs * 2u, res_co2 = (unsigned)a.res_co * 2u;
    const auto out_off = [&](int c) { return cp(c) * out_pitch2 + out_co2 + cqq(c) * 16u; };
    const auto res_off = [&](int c) { return cp(c) * res_pitch2 + res_co2 + cqq(c) * 16u; };
    const u32x4 res_rsrc = {sgpr((unsigned)(size_t)a.res), sgpr((unsigned)((size_t)a.res >> 32) & 0xffffu),
                            sgpr(RES && !(ABL & 32) ? 0xfffffff0u : 0u), sgpr(0x00020000u)};
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, OUT32 || (ABL & 2) ? 0u : 0xfffffff0u, 0x00020000);

    // ---- the tile pipeline's carried state ----
    floatx4 acc[3], pacc[3];            // tile g, tile g - 1
#pragma unroll
    for (int j = 0; j < 3; ++j) pacc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // output byte offset of the first pixel of tiles g - 1, g - 2, and the same in the shortcut tensor for tile g - 1; inv*:
    // all ones while there is no such tile yet (the offset is OR-ed out of range: no branch inside the tile loop)
    unsigned ob1 = 0, ob2 = 0, rb1 = 0, inv1 = 0xffffffffu, inv2 = 0xffffffffu;
    long pm1 = 0;                                    // OUT32: pixel index of tile g - 1 (-1: none)
    bool pv1 = false;
    u32x4 rres0 = {0, 0, 0, 0}, rres1 = {0, 0, 0, 0};   // shortcut chunks of tile g - 2
    half8 xf[3];
    unsigned A0 = 0, A1 = 0, A2 = 0;

    // fragment read of (tile ii of this wave, K step ks) into xf[slot]
    const auto read_frag = [&](int ii, int ks, int slot) {
        const int kh = ks <= 4 ? 0 : ks <= 8 ? 1 : 2;
        const unsigned base = ks == 4 ? (hi ? A1 - 32u : A0 + 256u) : ks == 13 ? A2 + 256u - (hi ? 32u : 0u) : kh == 0 ? A0 : kh == 1 ? A1 : A2;
        const int imm = (ks == 4 || ks == 13 ? 0 : 2 * (32 * ks - 144 * kh)) + ii * 16 * WS_PIX;
        if (ABL & 8) {
            asm volatile("" : "+v"(xf[slot]) : "v"(base));
            return;
        }
        xf[slot] = *(const __attribute__((address_space(3))) half8*)(size_t)(base + (unsigned)imm);
    };
    // B phase: value v (0..11) of tile g - 1
    float bv[4];
    floatx4 bj;
    const auto b_value = [&](int v) {
        const int j = v >> 2, e = v & 3;
        if (e == 0) bj = *(const floatx4*)(bias_p + j * 16 + cq);
        float x = pacc[j][e] + bj[e];    // the K steps in order on a zero accumulator, then the bias: the order of the kernel above
        if (ACT && !(ABL & 16)) x = silu_w(x);
        bv[e] = x;
        if (e == 3) {
            if (OUT32) {
                if (pv1) {
                    float* const o = a.out32 + (pm1 + px) * a.out_cs + a.out_co + j * 16 + cq;
                    if (RES) {
                        const __half* const rp = a.res + (pm1 + px) * a.res_cs + a.res_co + j * 16 + cq;
#pragma unroll
                        for (int t = 0; t < 4; ++t) bv[t] += __half2float(rp[t]);
                    }
                    *(float4*)o = make_float4(bv[0], bv[1], bv[2], bv[3]);
                }
            } else if (RES) {
                *(float4*)(sw + j * 64) = make_float4(bv[0], bv[1], bv[2], bv[3]);
            } else {
                union {
                    uint2 u;
                    _Float16 h[4];
                } o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o.h[t] = (_Float16)bv[t];
                *(uint2*)(sw + j * 32) = o.u;
            }
        }
    };
    // C phase: drain of tile g - 2, chunk 0 / 1: stage (+ shortcut) -> one 16-byte store
    u32x4 cda, cdb;
    const auto c_read = [&](int c) {
        if (OUT32) return;
        const unsigned char* const p = stage_rd(c);
        cda = *(const u32x4*)p;
        if (RES) cdb = *(const u32x4*)(p + 16);
    };
    const auto c_store = [&](int c) {
        if (OUT32) return;
        u32x4 o;
        if (RES) {
            const u32x4 fa = cda, fb = cdb, rr = c ? rres1 : rres0;
            union {
                u32x4 v;
                _Float16 h[8];
            } rh, oh;
            rh.v = rr;
            const float f[8] = {__uint_as_float(fa[0]), __uint_as_float(fa[1]), __uint_as_float(fa[2]), __uint_as_float(fa[3]),
                                __uint_as_float(fb[0]), __uint_as_float(fb[1]), __uint_as_float(fb[2]), __uint_as_float(fb[3])};
#pragma unroll
            for (int t = 0; t < 8; ++t) oh.h[t] = (_Float16)(f[t] + (float)rh.h[t]);
            o = oh.v;
        } else {
            o = cda;
        }
        const unsigned off = (ob2 + out_off(c)) | dead(c) | inv2;
        __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, off, 0, 0);
    };

    // One tile loop.  ii: this wave's tile within the step (compile time); T: its tiles per step; LAST: the step's last tile
    // (no prefetch across the barrier).  DPREV: row DMAs issued in the previous tile loop (after its shortcut loads).
    const auto tile_loop = [&](auto II, auto TT, const Pair& pair, unsigned ob0, unsigned rb0, long pm0) {
        constexpr int ii = decltype(II)::value, T = decltype(TT)::value;
        constexpr int prev = (ii + T - 1) % T;
        constexpr int DPREV = (NI - 4 * prev) < 0 ? 0 : (NI - 4 * prev) > 4 ? 4 : (NI - 4 * prev);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < WS_KSTEPS; ++ks) {
            // fragment reads two K steps ahead (into the next tile of the step where there is one); the three fragment
            // registers rotate over the step's K steps, not the tile's (14 is not a multiple of 3)
            constexpr int G0 = ii * WS_KSTEPS;
            if (ks + 2 < WS_KSTEPS)
                read_frag(ii, ks + 2, (G0 + ks + 2) % 3);
            else if (ii + 1 < T)
                read_frag(ii + 1, ks + 2 - WS_KSTEPS, (G0 + ks + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (ABL & 1)
                    asm volatile("" : "+v"(acc[j]) : "v"(wreg[ks][j]), "v"(xf[(G0 + ks) % 3]));
                else
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[ks][j], xf[(G0 + ks) % 3], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) {
                if (RES && !OUT32) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rres0), "+v"(rres1) : "n"(DPREV));
                asm volatile("" ::: "memory");   // the stage was written through other pointer types in the last tile loop
                c_read(0);
            }
            if (ks == 1) {   // one chunk at a time: the staged f32 values of both would cost 16 registers
                c_store(0);
                c_read(1);
            }
            if (ks == 2) c_store(1);
            if (ks == 3 && RES && !OUT32) {   // shortcut chunks of tile g - 1 (drained in the next tile loop)
                rres0 = ld16_asm(res_rsrc, (rb1 + res_off(0)) | inv1);
                rres1 = ld16_asm(res_rsrc, (rb1 + res_off(1)) | dead(1) | inv1);
            }
            if (ks >= 2) b_value(ks - 2);
            if (ks >= 5 && (ks & 1) && ks <= 11) {
                const int slot = ii * 4 + (ks - 5) / 2;
                if (slot < NI) issue_slot(pair, slot);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // tile g becomes tile g - 1
#pragma unroll
        for (int j = 0; j < 3; ++j) pacc[j] = acc[j];
        ob2 = ob1, inv2 = inv1;
        ob1 = ob0, inv1 = 0u;
        rb1 = rb0;
        pm1 = pm0;
        pv1 = true;
    };

    const auto run = [&](auto TT) {
        constexpr int T = decltype(TT)::value;
        constexpr int OPS = T * ((OUT32 ? 0 : 2) + (RES && !OUT32 ? 2 : 0)) + NI;          // vector-memory instructions per step
        constexpr int LASTD = (NI - 1) / 4;                                                  // tile loop of a step's last row DMA
        constexpr int AFTER = (T - 1 - LASTD) * ((OUT32 ? 0 : 2) + (RES && !OUT32 ? 2 : 0)); // ... and what the step issues behind it
        static_assert(OPS + AFTER <= 63, "vmcnt is 6 bits");
        for (int s = 0; s < steps; ++s) {
            // rows of this step: issued two steps ago (or by the prologue)
            if (s == 0)
                wait_vmw<NI>();
            else if (s == 1)
                wait_vmw<OPS>();
            else
                wait_vmw<OPS + AFTER>();
            if (!(ABL & 64)) __builtin_amdgcn_s_barrier();
            const Pair pair = pair_of(2 * s + 6);
            const int y = y_base + 2 * s + r;
            const long m_row = ((long)img_row0 + y) * WS_W + xh * 80 + t0 * 16;
            const unsigned vb0 = lds0 + ((2 * s + r + 0) % WP_SLOTS) * WS_ROW + lane_off;
            const unsigned vb1 = lds0 + ((2 * s + r + 1) % WP_SLOTS) * WS_ROW + lane_off;
            const unsigned vb2 = lds0 + ((2 * s + r + 2) % WP_SLOTS) * WS_ROW + lane_off;
            A0 = vb0, A1 = vb1, A2 = vb2;
            read_frag(0, 0, 0);
            read_frag(0, 1, 1);
            const unsigned ob = OUT32 ? 0u : (unsigned)(m_row * a.out_cs * 2);
            const unsigned rb = RES && !OUT32 ? (unsigned)(m_row * a.res_cs * 2) : 0u;
            const unsigned ostep = OUT32 ? 0u : (unsigned)(16 * a.out_cs * 2), rstep = RES && !OUT32 ? (unsigned)(16 * a.res_cs * 2) : 0u;
            tile_loop(std::integral_constant<int, 0>{}, TT, pair, ob, rb, m_row);
            if constexpr (T > 1) tile_loop(std::integral_constant<int, 1>{}, TT, pair, ob + ostep, rb + rstep, m_row + 16);
            if constexpr (T > 2) tile_loop(std::integral_constant<int, 2>{}, TT, pair, ob + 2 * ostep, rb + 2 * rstep, m_row + 32);
            if constexpr (T > 3) tile_loop(std::integral_constant<int, 3>{}, TT, pair, ob + 3 * ostep, rb + 3 * rstep, m_row + 48);
            if constexpr (T > 4) tile_loop(std::integral_constant<int, 4>{}, TT, pair, ob + 4 * ostep, rb + 4 * rstep, m_row + 64);
        }
        // ---- the strip's tail: B phase of the last tile, C phase of the last two (nothing left to hide them under) ----
        if (RES && !OUT32) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rres0), "+v"(rres1));
        asm volatile("" ::: "memory");
        c_read(0);
        c_store(0);
        c_read(1);
        c_store(1);
        asm volatile("" ::: "memory");
        if (RES && !OUT32) {
            rres0 = ld16_asm(res_rsrc, rb1 + res_off(0));
            rres1 = ld16_asm(res_rsrc, (rb1 + res_off(1)) | dead(1));
        }
#pragma unroll
        for (int v = 0; v < 12; ++v) b_value(v);
        ob2 = ob1, inv2 = inv1;
        if (RES && !OUT32) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rres0), "+v"(rres1));
        asm volatile("" ::: "memory");
        c_read(0);
        c_store(0);
        c_read(1);
        c_store(1);
        wait_vmw<0>();
    };
    if (NSPLIT == 1)
        run(std::integral_constant<int, 5>{});
    else if (sub == 0)
        run(std::integral_constant<int, 3>{});
    else
        run(std::integral_constant<int, 2>{});
}

// =====================================================================================================================
// conv_wsf_kernel (round 4): a whole C2f bottleneck of this shape in ONE launch -- out = x + SiLU(conv2(SiLU(conv1(x)))) --
// with the hidden tensor in LDS.  Why: the two launches move x in, h out, h in, x in again (the shortcut), y out = 5 tensors
// of 629 MB at 256 images; with a shortcut the second launch runs at 4.5 TB/s of real traffic, and the ablation of
// conv_wsp prices the stores of the first at 65 us and the input DMAs + shortcut loads of the second at ~150 us of a 885 us
// pair.  Fused: x in once, y out once.
//   * eight waves, two per SIMD: waves 0-3 hold the FIRST filter and compute hidden rows, waves 4-7 hold the SECOND and
//     compute output rows three steps behind; a step is one image row (ten 16-pixel tiles per convolution: the conv1 waves of
//     SIMDs 0-3 take 3, 3, 2, 2 of them, the conv2 waves 2, 2, 3, 3 -- five tiles per SIMD and step);
//   * LDS: a ring of six x rows (DMA two steps ahead; row s - 1 doubles as the shortcut of the output row of step s), a ring
//     of four hidden rows (written by conv1 in MFMA layout with the same zero edge pixels, read by conv2 as fragments), one
//     f16 output tile per conv2 wave, bias vectors: 163 200 of 163 840 bytes;
//   * hidden rows outside the image are zeros (conv2's padding), and a strip recomputes one hidden row above and below itself;
//   * a wave runs tile by tile -- 42 MFMAs, then the tile's bias + SiLU (+ shortcut from the x ring, + drain through the
//     stage for conv2) -- NOT software-pipelined as conv_wsp: with two filters' worth of code paths the carried accumulators
//     spilled (230 registers in one attempt), and conv_wsp's own measurements say the overlap that matters is the other
//     wave of the SIMD (one's epilogue beside the other's MFMAs), not the pipelining inside a wave;
//   * one f32 operation order per value, as the two launches: K steps in order on a zero accumulator, + bias, SiLU, h rounded
//     to f16 (what the unfused plan stores), ... + bias, SiLU, + (float)x, rounded to f16: bit-identical outputs.
constexpr int WF_XSLOTS = 6, WF_HSLOTS = 4;
constexpr int WF_STAGE = 1536;
constexpr int WF_LDS = (WF_XSLOTS + WF_HSLOTS) * WS_ROW + 4 * WF_STAGE + 1024 + 512;
static_assert(WF_LDS <= 160 * 1024, "the fused bottleneck's rings do not fit the LDS");

template <bool OUT32>
__global__ __launch_bounds__(512) void conv_wsf_kernel(const ConvArgs a, const int strip_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave & 3;            // SIMD
    const int role = wave >> 2;        // 0: conv1 (x -> hidden), 1: conv2 (hidden -> out)
    const bool three = role == 0 ? g < 2 : g >= 2;                       // three tiles per step (else two)
    const int t0 = role == 0 ? (g < 2 ? 3 * g : 6 + 2 * (g - 2)) : (g < 2 ? 2 * g : 4 + 3 * (g - 2));
    const int frow = lane & 15, kg = lane >> 4;
    const bool hi = kg >= 2;
    const int px = lane & 15, cq = (lane >> 4) * 4;

    const int strips = a.H / strip_rows;
    const int img = blockIdx.x / strips;
    const int y_base = (blockIdx.x % strips) * strip_rows;
    const int steps = strip_rows + 3;   // hidden rows y_base - 1 .. y_base + strip_rows at steps 0 .. strip_rows + 1; output row y_base + s - 3 at step s

    const u32x4 in_rsrc = {sgpr((unsigned)(size_t)a.in), sgpr((unsigned)((size_t)a.in >> 32) & 0xffffu), sgpr(a.in_bytes), sgpr(0x00020000u)};
    const unsigned hid0 = lds0 + WF_XSLOTS * WS_ROW;
    unsigned char* const stage_p = smem + (WF_XSLOTS + WF_HSLOTS) * WS_ROW + g * WF_STAGE;
    const unsigned scratch = sgpr(lds0 + (WF_XSLOTS + WF_HSLOTS) * WS_ROW + 4 * WF_STAGE);
    float* const bias_p = (float*)(smem + (WF_XSLOTS + WF_HSLOTS) * WS_ROW + 4 * WF_STAGE + 1024) + role * 64;

    for (int i = tid; i < (WF_XSLOTS + WF_HSLOTS) * 2 * (WS_PIX / 16); i += 512) {
        const int slot = i / (2 * (WS_PIX / 16));
        const int rem = i % (2 * (WS_PIX / 16));
        const int side = rem / (WS_PIX / 16), c16 = rem % (WS_PIX / 16);
        *(u32x4*)(smem + slot * WS_ROW + side * (WS_W + 1) * WS_PIX + c16 * 16) = u32x4{0, 0, 0, 0};
    }
    if (tid < WS_C) {
        float* const b0 = (float*)(smem + (WF_XSLOTS + WF_HSLOTS) * WS_ROW + 4 * WF_STAGE + 1024);
        b0[tid] = a.bias[tid];
        b0[64 + tid] = a.bias2[tid];
    }
    __syncthreads();

    half8 wreg[WS_KSTEPS][3];
    {
        const _Float16* const w = (const _Float16*)(role ? a.wt2 : a.wt);
#pragma unroll
        for (int ks = 0; ks < WS_KSTEPS; ++ks)
#pragma unroll
            for (int j = 0; j < 3; ++j) wreg[ks][j] = *(const half8*)(w + (size_t)(j * 16 + frow) * a.Kp + ks * 32 + kg * 8);
#pragma unroll
        for (int ks = 0; ks < WS_KSTEPS; ++ks)
#pragma unroll
            for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(wreg[ks][j]));
    }

    // ---- x rows: one row per step, 15 DMA instructions over 8 waves (slot q = wave + 8 j); relative row rx (0 = y_base - 2) in ring slot rx % 6
    const int img_row0 = img * a.H;
    const unsigned row_bytes = (unsigned)(WS_W * a.in_cs * 2);
    const auto issue_row = [&](int rx) {
        const int gy = y_base - 2 + rx;
        const unsigned live = (gy >= 0 && gy < a.H && rx <= strip_rows + 3) ? 0xffffffffu : 0u;
        const unsigned rowoff = (unsigned)(img_row0 + gy) * row_bytes;
        const unsigned slot = lds0 + (unsigned)(rx % WF_XSLOTS) * WS_ROW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = wave + 8 * j;                       // wave-uniform
            const unsigned q_ok = q < WS_DMA_ROW ? 0xffffffffu : 0u;
            unsigned ln = (unsigned)lane;
            asm volatile("" : "+v"(ln));                      // recomputed here, twice per step, instead of living in registers
            const unsigned c = (unsigned)(q % WS_DMA_ROW) * 64u + ln;
            const unsigned goff = ((c / 6u) * (unsigned)a.in_cs + (unsigned)a.in_co + (c % 6u) * 8u) * 2u;
            const unsigned off = (goff + rowoff) | ~(live & q_ok);
            const unsigned dst = (q_ok & (slot + (unsigned)(WS_PIX + (q % WS_DMA_ROW) * 1024))) | (~q_ok & scratch);
            dma16w(in_rsrc, sgpr(dst), off);
        }
    };
    issue_row(0);
    issue_row(1);
    issue_row(2);
    issue_row(3);

    const unsigned lane_off = (unsigned)(frow * WS_PIX + kg * 16 + t0 * 16 * WS_PIX);
    // this lane's values of a tile in a ring row / in the stage: pixel px (+ the zero edge pixel), channels j * 16 + cq ..
    const unsigned mf_off = (unsigned)((1 + t0 * 16 + px) * WS_PIX + cq * 2);
    unsigned char* const sw = stage_p + px * 96 + cq * 2;
    const int c0 = lane, c1 = (64 + lane) % 96;
    const unsigned pq = (unsigned)((c0 / 6) | ((c0 % 6) << 4) | ((c1 / 6) << 8) | ((c1 % 6) << 12) | (lane >= 32 ? 1 << 16 : 0));
    const auto pqv = [&]() {
        unsigned q = pq;
        asm volatile("" : "+v"(q));
        return q;
    };
    const auto cp = [&](int c) { return (pqv() >> (c ? 8 : 0)) & 15u; };
    const auto cqq = [&](int c) { return (pqv() >> (c ? 12 : 4)) & 15u; };
    const auto dead = [&](int c) { return c ? 0u - ((pqv() >> 16) & 1u) : 0u; };
    const unsigned out_pitch2 = (unsigned)a.out_cs * 2u, out_co2 = (unsigned)a.out_co * 2u;
    const auto out_off = [&](int c) { return cp(c) * out_pitch2 + out_co2 + cqq(c) * 16u; };
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, OUT32 ? 0u : 0xfffffff0u, 0x00020000);

    // fragment reads run PF K steps ahead of their MFMAs.  (A K step is only three MFMAs here; distances of 2, 3 and 4 K steps
    // measured the same 790-800 us per launch at 256 images -- the wave does not wait for LDS -- and 6 spills: PF = 2.)
    constexpr int PF = 2, XR = PF + 1;
    floatx4 acc[3];
    half8 xf[XR];
    unsigned A0 = 0, A1 = 0, A2 = 0;   // this lane's fragment address in the three ring rows a step reads
    const auto read_frag = [&](int ii, int ks, int slot) {
        const int kh = ks <= 4 ? 0 : ks <= 8 ? 1 : 2;
        const unsigned base = ks == 4 ? (hi ? A1 - 32u : A0 + 256u) : ks == 13 ? A2 + 256u - (hi ? 32u : 0u) : kh == 0 ? A0 : kh == 1 ? A1 : A2;
        const int imm = (ks == 4 || ks == 13 ? 0 : 2 * (32 * ks - 144 * kh)) + ii * 16 * WS_PIX;
        xf[slot] = *(const __attribute__((address_space(3))) half8*)(size_t)(base + (unsigned)imm);
    };
    // the 42 MFMAs of tile ii of this wave's step (T tiles): fragments two K steps ahead, into the next tile where there is one
    const auto k_loop = [&](auto II, auto TT) {
        constexpr int ii = decltype(II)::value, T = decltype(TT)::value;
        constexpr int G0 = ii * WS_KSTEPS;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < WS_KSTEPS; ++ks) {
            if (ks + PF < WS_KSTEPS)
                read_frag(ii, ks + PF, (G0 + ks + PF) % XR);
            else if (ii + 1 < T)
                read_frag(ii + 1, ks + PF - WS_KSTEPS, (G0 + ks + PF) % XR);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[ks][j], xf[(G0 + ks) % XR], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ================= conv1: hidden row of step s -> hidden ring (f16, what the two-launch plan stores) =================
    unsigned hrow = 0;           // LDS address of this lane's values of tile 0 of the wave in the hidden row being written
    float hmask = 1.f;           // 0: the hidden row lies outside the image (conv2 must read zeros there)
    const auto conv1_tile = [&](int tile) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const floatx4 bj = *(const floatx4*)(bias_p + j * 16 + cq);
            union {
                u32x2w v;
                _Float16 h[4];
            } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.h[e] = (_Float16)(silu_w(acc[j][e] + bj[e]) * hmask);
            *(__attribute__((address_space(3))) u32x2w*)(size_t)(hrow + (unsigned)(tile * 16 * WS_PIX + j * 32)) = o.v;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const auto conv1_step = [&](auto TT) {
        constexpr int T = decltype(TT)::value;
#pragma unroll
        for (int p = 0; p < PF; ++p) read_frag(0, p, p);
        k_loop(std::integral_constant<int, 0>{}, TT);
        conv1_tile(0);
        k_loop(std::integral_constant<int, 1>{}, TT);
        conv1_tile(1);
        if constexpr (T > 2) {
            k_loop(std::integral_constant<int, 2>{}, TT);
            conv1_tile(2);
        }
    };

    // ================= conv2: output row of step s: + bias, SiLU, + x (from the x ring), f16, through the stage =================
    const auto conv2_tile = [&](int tile, unsigned xres, unsigned ob, long pm) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const floatx4 bj = *(const floatx4*)(bias_p + j * 16 + cq);
            union {
                u32x2w v;
                _Float16 h[4];
            } rr, o;
            rr.v = *(const __attribute__((address_space(3))) u32x2w*)(size_t)(xres + (unsigned)(tile * 16 * WS_PIX + j * 32));
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_w(acc[j][e] + bj[e]) + (float)rr.h[e];
            if (OUT32) {
                *(float4*)(a.out32 + (pm + tile * 16 + px) * a.out_cs + a.out_co + j * 16 + cq) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o.h[e] = (_Float16)v[e];
                *(u32x2w*)(sw + j * 32) = o.v;
            }
            __builtin_amdgcn_sched_barrier(0);   // one channel tile at a time: the scheduler otherwise keeps all twelve values live
        }
        if (OUT32) return;
        asm volatile("" ::: "memory");
        const unsigned obt = ob + (unsigned)tile * (unsigned)(16 * a.out_cs * 2);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const u32x4 d = *(const u32x4*)(stage_p + cp(c) * 96u + cqq(c) * 16u);
            __builtin_amdgcn_raw_buffer_store_b128(d, out_rsrc, (obt + out_off(c)) | dead(c), 0, 0);
        }
        asm volatile("" ::: "memory");
    };
    const auto conv2_step = [&](auto TT, unsigned xres, unsigned ob, long pm) {
        constexpr int T = decltype(TT)::value;
#pragma unroll
        for (int p = 0; p < PF; ++p) read_frag(0, p, p);
        k_loop(std::integral_constant<int, 0>{}, TT);
        conv2_tile(0, xres, ob, pm);
        k_loop(std::integral_constant<int, 1>{}, TT);
        conv2_tile(1, xres, ob, pm);
        if constexpr (T > 2) {
            k_loop(std::integral_constant<int, 2>{}, TT);
            conv2_tile(2, xres, ob, pm);
        }
    };

    for (int s = 0; s < steps; ++s) {
        // the x row conv1 reads last in this step (rx = s + 2) was issued at the start of step s - 2 (or by the prologue); the
        // instructions issued since: the two row DMAs of step s - 1 and, on the conv2 waves, two stores per tile of steps
        // s - 2 and s - 1 (they store from step 3 on)
        if (role == 0 || s <= 3)
            wait_vmw<2>();
        else if (s == 4) {
            if (three)
                wait_vmw<2 + 6>();
            else
                wait_vmw<2 + 4>();
        } else {
            if (three)
                wait_vmw<2 + 12>();
            else
                wait_vmw<2 + 8>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's hidden-row writes (and ring reads) of the step before
        __builtin_amdgcn_s_barrier();
        issue_row(s + 4);   // into the slot of row s - 2, the shortcut row of step s - 1
        // (starting the conv2 waves 256-768 cycles late, so that the two waves of a SIMD are half a tile out of phase: no change)
        // (opaque copies: the optimiser otherwise keeps "lane offset + slot base" for every slot and tile as induction variables
        // across the step loop -- twenty-odd registers this kernel does not have)
        unsigned lane_off_s = lane_off, mf_off_s = mf_off;
        asm volatile("" : "+v"(lane_off_s), "+v"(mf_off_s));
        if (role == 0) {
            if (s <= strip_rows + 1) {
                const int hr = y_base - 1 + s;
                hmask = hr >= 0 && hr < a.H ? 1.f : 0.f;
                A0 = lds0 + (unsigned)((s + 0) % WF_XSLOTS) * WS_ROW + lane_off_s;
                A1 = lds0 + (unsigned)((s + 1) % WF_XSLOTS) * WS_ROW + lane_off_s;
                A2 = lds0 + (unsigned)((s + 2) % WF_XSLOTS) * WS_ROW + lane_off_s;
                hrow = hid0 + (unsigned)(s % WF_HSLOTS) * WS_ROW + mf_off_s;
                if (three)
                    conv1_step(std::integral_constant<int, 3>{});
                else
                    conv1_step(std::integral_constant<int, 2>{});
            }
        } else if (s >= 3) {
            const int y = y_base + s - 3;
            A0 = hid0 + (unsigned)((s - 3) % WF_HSLOTS) * WS_ROW + lane_off_s;
            A1 = hid0 + (unsigned)((s - 2) % WF_HSLOTS) * WS_ROW + lane_off_s;
            A2 = hid0 + (unsigned)((s - 1) % WF_HSLOTS) * WS_ROW + lane_off_s;
            const unsigned xres = lds0 + (unsigned)((s - 1) % WF_XSLOTS) * WS_ROW + mf_off_s;
            const long m_row = ((long)img_row0 + y) * WS_W + t0 * 16;
            const unsigned ob = OUT32 ? 0u : (unsigned)(m_row * a.out_cs * 2);
            if (three)
                conv2_step(std::integral_constant<int, 3>{}, xres, ob, m_row);
            else
                conv2_step(std::integral_constant<int, 2>{}, xres, ob, m_row);
        }
    }
    wait_vmw<0>();
}

// variant = strip height x wave layout (NJ = 1 for ids 0..5, NJ = 3 for ids 6..11); 12..: the pipelined kernel
const int kWsStripRows[] = {40, 20, 10, 8, 4, 2};
constexpr int kNumStrips = sizeof(kWsStripRows) / sizeof(kWsStripRows[0]);
constexpr int kNumWsOld = 2 * kNumStrips;
struct WspVariant {
    int strip_rows, nsplit;
};
// ids 12..16.  (Measured at 256 images, us per launch without / with a shortcut: kernel above 442 / 533; two waves per SIMD,
// strips of 160 / 80 / 40 / 20 / 10 rows 397 / 508, 413 / 520, 450 / 543, 481 / 588, 575 / 684; one wave per SIMD 453 / 573:
// the tile-outer order alone buys nothing, the second wave per SIMD does.)
const WspVariant kWsp[] = {{160, 2}, {80, 2}, {40, 2}, {20, 2}, {160, 1}};
constexpr int kNumWsp = sizeof(kWsp) / sizeof(kWsp[0]);
constexpr int kNumWs = kNumWsOld + kNumWsp;

const int kWsfStripRows[] = {160, 80, 40, 20};   // conv_wsf variants
constexpr int kNumWsf = sizeof(kWsfStripRows) / sizeof(kWsfStripRows[0]);

}  // namespace

int conv_wsf_num_variants() { return kNumWsf; }

bool conv_wsf_supported(const ConvArgs& a, int variant) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || !a.act || a.pre || a.in_slab_c || a.out_slab_c) return false;
    if (a.Cin != WS_C || a.Cout_pad != WS_C || a.W != WS_W || a.Wo != a.W || a.Ho != a.H) return false;
    if (a.Kp < WS_KSTEPS * 32 || (!a.out32 && !a.out) || !a.wt2 || !a.bias2) return false;
    if (variant < 0) return a.H % 2 == 0;
    return variant < kNumWsf && a.H % kWsfStripRows[variant] == 0;
}

void launch_conv_wsf(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int variant) {
    if (variant < 0 || variant >= kNumWsf) fail(RMR_ERR_INVALID_ARGUMENT, "conv_wsf: variant %d out of range", variant);
    if (!conv_wsf_supported(a, variant)) fail(RMR_ERR_LOGIC, "conv_wsf: layer pair not supported by variant %d", variant);
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 8 || a.out_co % 8) fail(RMR_ERR_LOGIC, "conv_wsf: misaligned view");
    if (a.in_bytes == 0 || a.in_bytes > 0xf0000000ull) fail(RMR_ERR_LOGIC, "conv_wsf: input view size not set or larger than 3.75 GiB");
    if (!a.out32 && (double)a.M * a.out_cs * 2 >= 4.0e9) fail(RMR_ERR_LOGIC, "conv_wsf: output view of 4 GB or more (32-bit store offsets)");
    // The input is read twice -- as the first convolution's operand (with halo rows that other workgroups own) and as the
    // shortcut -- while the output is written: an output view on top of the input view (same bytes, meeting channel ranges)
    // would race across the strips' halos.  (The two-launch path tolerates out == res; this one does not.)
    if (!a.out32) {
        const char *ib = (const char*)a.in, *ie = ib + (size_t)a.N * a.H * a.W * a.in_cs * 2;
        const char *ob = (const char*)a.out, *oe = ob + (size_t)a.M * a.out_cs * 2;
        const bool bytes_meet = ob < ie && ib < oe;
        const bool same_pitch = a.in_cs == a.out_cs && ((ob - ib) % ((long)a.in_cs * 2)) == 0;
        const bool channels_meet = !same_pitch || (a.out_co < a.in_co + a.Cin && a.in_co < a.out_co + a.Cout_pad);
        if (bytes_meet && channels_meet) fail(RMR_ERR_LOGIC, "conv_wsf: the output view overlaps the input view (the fused bottleneck cannot run in place)");
    }
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)conv_wsf_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_wsf_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int sr = kWsfStripRows[variant];
    const int grid = a.N * (a.H / sr);
    // both convolutions; algorithmic bytes: x in, y out, two filters (the hidden tensor and the shortcut read never reach HBM)
    const double flops = a.flops > 0 ? a.flops : 2.0 * 2.0 * a.M * (double)a.Cout_pad * a.K;
    const double bytes = 2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)a.M * a.Cout_pad + 2.0 * a.Cout_pad * a.K);
    static const bool per_layer = std::getenv("RMR_PROFILE_LAYERS") != nullptr;
    static std::mutex name_mu;
    static std::map<std::string, std::string> names;
    const char* pname = "conv_igemm_f16";
    if (per_layer && ctx.prof.on) {
        char buf[64];
        snprintf(buf, sizeof(buf), "conv n%d M%d N%d K%d k%d s%d x2 b%d", a.N, a.M, a.Cout_pad, a.K, a.KH, a.stride, variant);
        std::lock_guard<std::mutex> lk(name_mu);
        pname = names.emplace(buf, buf).first->second.c_str();
    }
    ProfScope ps(ctx.prof, stream, pname, flops, bytes);
    if (a.out32)
        conv_wsf_kernel<true><<<grid, 512, WF_LDS, stream>>>(a, sr);
    else
        conv_wsf_kernel<false><<<grid, 512, WF_LDS, stream>>>(a, sr);
    RMR_HIP(hipGetLastError());
}

int conv_ws_num_variants() { return kNumWs; }

bool conv_ws_supported(const ConvArgs& a, int variant) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1) return false;
    if (a.Cin != WS_C || a.Cout_pad != WS_C || a.W != WS_W || a.Wo != a.W || a.Ho != a.H) return false;
    if (a.Kp < WS_KSTEPS * 32 || (!a.out32 && !a.out)) return false;
    if (variant < 0) return a.H % 2 == 0;
    if (variant >= kNumWsOld) return variant < kNumWs && a.H % kWsp[variant - kNumWsOld].strip_rows == 0;
    return a.H % kWsStripRows[variant % kNumStrips] == 0;
}

void launch_conv_ws(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int variant) {
    if (variant < 0 || variant >= kNumWs) fail(RMR_ERR_INVALID_ARGUMENT, "conv_ws: variant %d out of range", variant);
    if (!conv_ws_supported(a, variant)) fail(RMR_ERR_LOGIC, "conv_ws: layer not supported by variant %d", variant);
    if (a.in_cs % 8 || a.in_co % 8 || a.out_cs % 8 || a.out_co % 8 || a.res_cs % 8 || a.res_co % 8)
        fail(RMR_ERR_LOGIC, "conv_ws: misaligned view");
    if (a.in_bytes == 0 || a.in_bytes > 0xf0000000ull) fail(RMR_ERR_LOGIC, "conv_ws: input view size not set or larger than 3.75 GiB");
    if (!a.out32 && (double)a.M * a.out_cs * 2 >= 4.0e9) fail(RMR_ERR_LOGIC, "conv_ws: output view of 4 GB or more (32-bit store offsets)");
    // the shortcut is addressed with 32-bit byte offsets as well, and may be a slice of a WIDER buffer than the output
    if (a.res && (double)a.M * a.res_cs * 2 >= 4.0e9) fail(RMR_ERR_LOGIC, "conv_ws: shortcut view of 4 GB or more (32-bit load offsets)");
    using Kern = void (*)(const ConvArgs, int);
#define WS_KERNELS(NJ)                                                                                   \
    {conv_ws_kernel<false, false, false, NJ>, conv_ws_kernel<false, false, true, NJ>,                    \
     conv_ws_kernel<false, true, false, NJ>,  conv_ws_kernel<false, true, true, NJ>,                     \
     conv_ws_kernel<true, false, false, NJ>,  conv_ws_kernel<true, false, true, NJ>,                     \
     conv_ws_kernel<true, true, false, NJ>,   conv_ws_kernel<true, true, true, NJ>}
    static const Kern kernels[2][8] = {WS_KERNELS(1), WS_KERNELS(3)};
#undef WS_KERNELS
#define WSP_KERNELS(NS)                                                                                    \
    {conv_wsp_kernel<false, false, false, NS>, conv_wsp_kernel<false, false, true, NS>,                    \
     conv_wsp_kernel<false, true, false, NS>,  conv_wsp_kernel<false, true, true, NS>,                     \
     conv_wsp_kernel<true, false, false, NS>,  conv_wsp_kernel<true, false, true, NS>,                     \
     conv_wsp_kernel<true, true, false, NS>,   conv_wsp_kernel<true, true, true, NS>}
    static const Kern pkernels[2][8] = {WSP_KERNELS(1), WSP_KERNELS(2)};
#undef WSP_KERNELS
#ifdef RMR_WSP_ABLATE
#define WSP_ABL(M) {M, conv_wsp_kernel<true, false, false, 2, M>, conv_wsp_kernel<true, true, false, 2, M>}
    struct AblK {
        int mask;
        Kern plain, res;
    };
    static const AblK abl[] = {WSP_ABL(1), WSP_ABL(2), WSP_ABL(4), WSP_ABL(8), WSP_ABL(16), WSP_ABL(32), WSP_ABL(64), WSP_ABL(38), WSP_ABL(54),
                               WSP_ABL(17), WSP_ABL(63), WSP_ABL(9)};
#undef WSP_ABL
#endif
    static std::once_flag once;
    std::call_once(once, [] {
        for (const auto& row : kernels)
            for (Kern k : row)
                (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (const auto& row : pkernels)
            for (Kern k : row)
                (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const bool piped = variant >= kNumWsOld;
    const int nj = piped ? 1 : variant < kNumStrips ? 1 : 3;
    const int nsplit = piped ? kWsp[variant - kNumWsOld].nsplit : 1;
    const int kidx = (a.act ? 4 : 0) + (a.res ? 2 : 0) + (a.out32 ? 1 : 0);
    Kern kernel = piped ? pkernels[nsplit - 1][kidx] : kernels[nj == 1 ? 0 : 1][kidx];
#ifdef RMR_WSP_ABLATE
    if (const char* e = std::getenv("RMR_WSP_ABLATE")) {
        const int m = std::atoi(e);
        if (m && piped && nsplit == 2 && a.act && !a.out32) {
            kernel = nullptr;
            for (const AblK& k : abl)
                if (k.mask == m) kernel = a.res ? k.res : k.plain;
            if (!kernel) fail(RMR_ERR_INVALID_ARGUMENT, "conv_wsp: ablation mask %d is not compiled in", m);
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
    }
#endif
    const int sr = piped ? kWsp[variant - kNumWsOld].strip_rows : kWsStripRows[variant % kNumStrips];
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
    if (piped)
        kernel<<<grid, 256 * nsplit, wp_lds(4 * nsplit), stream>>>(a, sr);
    else
        kernel<<<grid, 256 * nj, ws_lds(nj), stream>>>(a, sr);
    RMR_HIP(hipGetLastError());
}

}  // namespace rmr
