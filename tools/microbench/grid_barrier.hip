// What a grid-wide barrier costs INSIDE a launch on this chip, for the grid shapes of the small-batch convolutions (conv_sb.hip):
// the price a persistent "one launch per network stage" kernel pays between two layers, to set against the kernel boundary it
// replaces (VERDICT r04 item 1 and its kill criterion).
//
// One launch runs `phases` phases; a phase = every workgroup writes `bytes` of "activations" (16-byte stores), then the grid
// barrier, then reads `bytes` written by ANOTHER workgroup (the next layer's input always comes from other CUs) and checks them.
// Barrier forms:
//   flat : one counter (agent-scope atomic add; lane 0 release fence before, acquire fence after; relaxed polling with s_sleep)
//   xcd  : per-XCD counters, the last arriver of an XCD adds to a top counter, every workgroup polls its XCD's generation word
// and the same work as separate launches (one launch per phase, back to back on one stream) as the baseline.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/microbench/grid_barrier.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Sync {
    unsigned flat;          // arrivals, monotonic over the launch
    unsigned pad0[31];
    unsigned top;           // XCDs arrived, monotonic
    unsigned pad1[31];
    unsigned xcd_cnt[8 * 32];   // arrivals per XCD (one 128-byte line each)
    unsigned xcd_gen[8 * 32];   // generation per XCD
    unsigned fail;
};

__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__device__ __forceinline__ void grid_barrier(Sync* s, unsigned phase, unsigned n_wg, unsigned wg_per_xcd_of_mine, unsigned xcd) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 0) {
            __hip_atomic_fetch_add(&s->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (phase + 1) * n_wg;
            unsigned spins = 0;
            while (ld_relaxed(&s->flat) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) {
                    s->fail = 1;
                    break;
                }
            }
        } else {
            const unsigned t = __hip_atomic_fetch_add(&s->xcd_cnt[xcd * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == (phase + 1) * wg_per_xcd_of_mine) {   // the XCD's last arriver reports the XCD
                const unsigned tt = __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tt + 1 == (phase + 1) * 8u) {              // the last XCD releases everybody
                    for (int x = 0; x < 8; ++x) __hip_atomic_store(&s->xcd_gen[x * 32], phase + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            unsigned spins = 0;
            while (ld_relaxed(&s->xcd_gen[xcd * 32]) < phase + 1) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) {
                    s->fail = 1;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// the work of one phase: write my slab, (barrier or kernel boundary), read the slab of workgroup (b + 37) % n and check it
__device__ __forceinline__ void write_slab(u32x4* buf, int bytes, unsigned phase, unsigned b) {
    for (int i = threadIdx.x; i < bytes / 16; i += blockDim.x) buf[(size_t)b * (bytes / 16) + i] = u32x4{phase, b, (unsigned)i, phase ^ b};
}
__device__ __forceinline__ void read_slab(const u32x4* buf, int bytes, unsigned phase, unsigned b, unsigned n, Sync* s) {
    const unsigned o = (b + 37u) % n;
    unsigned bad = 0;
    for (int i = threadIdx.x; i < bytes / 16; i += blockDim.x) {
        const u32x4 v = buf[(size_t)o * (bytes / 16) + i];
        bad |= (v[0] != phase) | (v[1] != o) | (v[2] != (unsigned)i);
    }
    if (bad) s->fail = 2;
}

template <int MODE>
__global__ __launch_bounds__(512) void persistent(u32x4* buf0, u32x4* buf1, int bytes, int phases, Sync* s) {
    const unsigned b = blockIdx.x, n = gridDim.x, xcd = b & 7;
    const unsigned mine = n / 8 + (xcd < n % 8 ? 1 : 0);
    for (int p = 0; p < phases; ++p) {
        u32x4* const buf = (p & 1) ? buf1 : buf0;
        write_slab(buf, bytes, p, b);
        grid_barrier<MODE>(s, p, n, mine, xcd);
        read_slab(buf, bytes, p, b, n, s);
    }
}

__global__ __launch_bounds__(512) void one_phase(u32x4* buf, const u32x4* prev, int bytes, int p, Sync* s) {
    const unsigned b = blockIdx.x, n = gridDim.x;
    if (p > 0) read_slab(prev, bytes, p - 1, b, n, s);
    write_slab(buf, bytes, p, b);
}

int main() {
    const int phases = 40;
    Sync* s;
    (void)hipMalloc(&s, sizeof(Sync));
    u32x4 *b0, *b1;
    (void)hipMalloc(&b0, 64 << 20);
    (void)hipMalloc(&b1, 64 << 20);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    printf("%d phases; a phase writes `bytes` per workgroup and reads another workgroup's\n", phases);
    for (int grid : {117, 150, 256, 512})
        for (int threads : {256, 448})
            for (int bytes : {0, 4096, 16384}) {
                float ms[3] = {0, 0, 0};
                int fail[3] = {0, 0, 0};
                for (int mode = 0; mode < 3; ++mode) {
                    float best = 1e30f;
                    for (int rep = 0; rep < 5; ++rep) {
                        (void)hipMemset(s, 0, sizeof(Sync));
                        (void)hipDeviceSynchronize();
                        (void)hipEventRecord(e0);
                        if (mode == 0)
                            persistent<0><<<grid, threads>>>(b0, b1, bytes, phases, s);
                        else if (mode == 1)
                            persistent<1><<<grid, threads>>>(b0, b1, bytes, phases, s);
                        else
                            for (int p = 0; p < phases; ++p) one_phase<<<grid, threads>>>((p & 1) ? b1 : b0, (p & 1) ? b0 : b1, bytes, p, s);
                        (void)hipEventRecord(e1);
                        (void)hipEventSynchronize(e1);
                        float t;
                        (void)hipEventElapsedTime(&t, e0, e1);
                        if (rep > 0 && t < best) best = t;
                        Sync h;
                        (void)hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost);
                        fail[mode] |= (int)h.fail;
                    }
                    ms[mode] = best;
                }
                printf("grid %3d x %3d threads, %5d B per workgroup: per phase  flat barrier %5.2f us  xcd barrier %5.2f us  separate launches %5.2f us%s\n", grid,
                       threads, bytes, ms[0] * 1e3 / phases, ms[1] * 1e3 / phases, ms[2] * 1e3 / phases, (fail[0] | fail[1] | fail[2]) ? "   (FAILED CHECK)" : "");
            }
    return 0;
}
