// locator.hip -- the LiDAR Locator on the GPU (src/locate/locate.cpp:37-350).
//
// State lives in HBM for the life of the object: the running-max background depth image,
// a ring of the last queue_size depth images, the foreground ("diff") depth image and a
// 64-bit key image used to make the per-pixel scatter deterministic.
//
//   update()   loc_scatter   one thread per point: filter, extrinsic transform, pinhole
//                            projection, then two atomics per point -- u64 atomicMax of
//                            ((index+1)<<32 | depth bits) => "highest point index wins"
//                            (Q13), and int atomicMax on the background (valid because the
//                            background is >= 0 and only d > background updates it);
//              loc_diff      one thread per pixel: key -> ring slot, clear key, walk the
//                            ring oldest -> newest (Q14), write the foreground image;
//   cluster()  fg_count / fg_scan / fg_compact : row-major ordered compaction of the
//                            non-zero foreground pixels + cameraToLidar (Q16);
//              cc_fused      Euclidean clustering = connected components of
//                            "squared distance < tolerance^2" by lock-free union-find
//                            (hook the larger root under the smaller) in ONE workgroup
//                            with the forest in LDS, size filter, ids ordered by
//                            (size desc, lowest member index) (Q17);
//   search()   loc_search    one workgroup per robot over the compact foreground list:
//                            LDS bucket histogram, first-max winner (-1 wins ties, Q18),
//                            f64 tree-sum centroid, lidar->world, mm -> m.
//
// All of it is HBM/latency-bound scatter/scan work: coalesced 4-byte streams, no LDS tiling
// of the images, and launch counts kept small.  Pixel binning and every comparison is
// bit-exact with the oracle (same f32 op order, -ffp-contract=off); only the centroid sum
// differs in association (tolerance 1e-3 m per BASELINE.json).
#include "locator.h"

#include <cmath>
#include <mutex>

namespace rmr {

constexpr int MAX_QUEUE = 16;
struct RingOrder {
    int n;
    int slot[MAX_QUEUE];
    int newest;
};

// ---- shared f32 geometry, same operation order as cv::Matx products ---------------------
__host__ __device__ inline void mat3_vec(const float* M, const float* v, float* o) {
    for (int i = 0; i < 3; ++i) {
        float s = 0;
        for (int k = 0; k < 3; ++k) s += M[i * 3 + k] * v[k];
        o[i] = s;
    }
}

// locate.cpp:73-81
__host__ __device__ inline void lidar_to_camera(const LocParams& P, const float* p, float* uvd) {
    const float v[4] = {p[0], p[1], p[2], 1.0f};
    float c4[3];
    for (int i = 0; i < 3; ++i) {
        float s = 0;
        for (int k = 0; k < 4; ++k) s += P.L2C[i * 4 + k] * v[k];
        c4[i] = s;
    }
    float c[3];
    mat3_vec(P.K, c4, c);
    uvd[0] = c[0] * P.zoom / c[2];
    uvd[1] = c[1] * P.zoom / c[2];
    uvd[2] = c[2];
}

// locate.cpp:54-61 (Q16): R * ((Kinv * d) * [u/z, v/z, 1] + t)
__host__ __device__ inline void camera_to_lidar(const LocParams& P, const float* uvd, float* out) {
    const float cam[3] = {uvd[0] / P.zoom, uvd[1] / P.zoom, 1.0f};
    float Ks[9];
    for (int i = 0; i < 9; ++i) Ks[i] = P.Kinv[i] * uvd[2];
    float q[3];
    mat3_vec(Ks, cam, q);
    for (int i = 0; i < 3; ++i) q[i] = q[i] + P.t[i];
    mat3_vec(P.R, q, out);
}

// locate.cpp:37-42 with the constant product C2W * L2C formed once (same f32 result)
__host__ __device__ inline void lidar_to_world(const LocParams& P, const float* p, float* out) {
    const float v[4] = {p[0], p[1], p[2], 1.0f};
    for (int i = 0; i < 3; ++i) {
        float s = 0;
        for (int k = 0; k < 4; ++k) s += P.L2W[i * 4 + k] * v[k];
        out[i] = s;
    }
}

// ---- update() ----------------------------------------------------------------------------

// locate.cpp:173-193
__global__ __launch_bounds__(256) void loc_scatter(LocParams P, const char* __restrict__ xyz,
                                                   int n, int stride_bytes,
                                                   unsigned long long* __restrict__ key,
                                                   float* __restrict__ bg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* pt = (const float*)(xyz + (size_t)i * stride_bytes);
    const float p[3] = {pt[0], pt[1], pt[2]};
    if (p[0] == 0 && p[1] == 0 && p[2] == 0) return;  // locate.cpp:176
    if (p[0] > P.max_distance) return;                // locate.cpp:179
    float uvd[3];
    lidar_to_camera(P, p, uvd);
    const float u = uvd[0], v = uvd[1], d = uvd[2];
    // locate.cpp:184-187 with Q15: u == Wz / v == Hz (and NaN) are out of range here
    if (!(u >= 0 && u < (float)P.wz && v >= 0 && v < (float)P.hz)) return;
    const size_t idx = (size_t)(int)v * P.wz + (int)u;
    // background = running max (locate.cpp:188-191); it never goes below +0, so only
    // positive depths can raise it and positive floats order like their int bits
    if (d > 0) atomicMax((int*)&bg[idx], __float_as_int(d));
    // depth = last writer in point order (locate.cpp:192, Q13)
    const unsigned long long k =
        ((unsigned long long)(unsigned)(i + 1) << 32) | (unsigned)__float_as_uint(d);
    atomicMax(&key[idx], k);
}

// locate.cpp:195-219
__global__ __launch_bounds__(256) void loc_diff(LocParams P, RingOrder order,
                                                unsigned long long* __restrict__ key,
                                                const float* __restrict__ bg,
                                                float* __restrict__ ring,
                                                float* __restrict__ diff, size_t npx) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npx) return;
    const unsigned long long k = key[p];
    const float cur = k ? __uint_as_float((unsigned)(k & 0xffffffffull)) : 0.0f;
    if (k) key[p] = 0;
    ring[(size_t)order.newest * npx + p] = cur;
    const float b = bg[p];
    float out = 0.0f;
    for (int q = 0; q < order.n; ++q) {
        const int s = order.slot[q];
        const float value = (s == order.newest) ? cur : ring[(size_t)s * npx + p];
        if (value == 0) continue;
        const float df = b - value;
        if (df >= P.min_diff && df <= P.max_diff) out = value;
    }
    diff[p] = out;
}

// ---- update() of a batch of consecutive frames -------------------------------------------------
// The updates of a stream are ordered (the background is a running max, the ring is the stream's history), but the
// order only matters PER PIXEL.  So a batch needs two launches instead of two per frame:
//   loc_scatter_batch  every point of every frame at once, into per-frame images: key[f] (the u64 "highest point index
//                      wins" key of loc_scatter) and fmax[f] (the frame's largest positive depth per pixel, which is
//                      what the frame contributes to the running-max background);
//   loc_walk_batch     one thread per pixel walks the frames in order with the background and the ring of the last Q
//                      depths in registers: exactly the sequence loc_scatter; loc_diff; loc_scatter; loc_diff ... sees
//                      at that pixel, so the foreground images, the background and the ring are the same bits.
struct CloudTable {
    const char* xyz;   // device address of the frame's points (nullptr / n == 0: locate.cpp:160-171, nothing queued)
    int n;
    int pad;
};

__global__ __launch_bounds__(256) void loc_scatter_batch(LocParams P, const CloudTable* __restrict__ clouds, int stride_bytes,
                                                         unsigned long long* __restrict__ key, int* __restrict__ fmax, size_t npx) {
    const CloudTable c = clouds[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= c.n) return;
    const float* pt = (const float*)(c.xyz + (size_t)i * stride_bytes);
    const float p[3] = {pt[0], pt[1], pt[2]};
    if (p[0] == 0 && p[1] == 0 && p[2] == 0) return;  // locate.cpp:176
    if (p[0] > P.max_distance) return;                // locate.cpp:179
    float uvd[3];
    lidar_to_camera(P, p, uvd);
    const float u = uvd[0], v = uvd[1], d = uvd[2];
    if (!(u >= 0 && u < (float)P.wz && v >= 0 && v < (float)P.hz)) return;
    const size_t idx = (size_t)blockIdx.y * npx + (size_t)(int)v * P.wz + (int)u;
    if (d > 0) atomicMax(&fmax[idx], __float_as_int(d));
    const unsigned long long k = ((unsigned long long)(unsigned)(i + 1) << 32) | (unsigned)__float_as_uint(d);
    atomicMax(&key[idx], k);
}

// ring_in: the stream's ring before the batch, oldest first at slots (head + q) % Q; after the batch the ring is stored
// oldest first from slot 0 (the host sets head = 0).  len0: images in the ring before the batch.
template <int Q>
__global__ __launch_bounds__(256) void loc_walk_batch(LocParams P, const CloudTable* __restrict__ clouds, int n_frames,
                                                      unsigned long long* __restrict__ key, int* __restrict__ fmax,
                                                      float* __restrict__ bg, float* __restrict__ ring, int head, int len0,
                                                      float* __restrict__ diff, size_t npx) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npx) return;
    float r[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) r[q] = q < len0 ? ring[(size_t)((head + q) % Q) * npx + p] : 0.0f;
    float b = bg[p];
    int len = len0;
    for (int f = 0; f < n_frames; ++f) {
        const size_t at = (size_t)f * npx + p;
        if (clouds[f].n <= 0 || !clouds[f].xyz) {   // locate.cpp:160-171: foreground cleared, nothing queued
            diff[at] = 0.0f;
            continue;
        }
        const unsigned long long k = key[at];
        const int m = fmax[at];
        const float cur = k ? __uint_as_float((unsigned)(k & 0xffffffffull)) : 0.0f;
        if (k) key[at] = 0;
        if (m) {
            fmax[at] = 0;
            b = fmaxf(b, __int_as_float(m));   // both >= +0: the int atomicMax of loc_scatter on the background itself
        }
        if (len < Q) {   // uniform: push_back ...
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (q == len) r[q] = cur;
            ++len;
        } else {         // ... pop_front when over queue_size (locate.cpp:195-198)
#pragma unroll
            for (int q = 0; q + 1 < Q; ++q) r[q] = r[q + 1];
            r[Q - 1] = cur;
        }
        float out = 0.0f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float value = r[q];
            if (q < len && value != 0) {
                const float df = b - value;
                if (df >= P.min_diff && df <= P.max_diff) out = value;
            }
        }
        diff[at] = out;
    }
    bg[p] = b;
#pragma unroll
    for (int q = 0; q < Q; ++q)
        if (q < len) ring[(size_t)q * npx + p] = r[q];
}

// ---- cluster(): ordered compaction ---------------------------------------------------------

constexpr int FG_PX_PER_BLOCK = 1024;  // 256 threads x 4 consecutive pixels

// The cluster() kernels serve one frame or a batch of frames in one launch: blockIdx.y (grids over pixels or
// points) or blockIdx.x (one workgroup per frame) is the frame; per-frame scratch lies frame-major, the
// products go to consecutive FrameSlots (slot_int_stride ints / slot_f_stride floats apart).
__global__ __launch_bounds__(256) void fg_count(const float* __restrict__ diff, size_t npx,
                                                int* __restrict__ blk_count) {
    __shared__ int wsum[4];
    diff += (size_t)blockIdx.y * npx;
    blk_count += (size_t)blockIdx.y * gridDim.x;
    const size_t base = (size_t)blockIdx.x * FG_PX_PER_BLOCK + (size_t)threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (base + j < npx && diff[base + j] != 0) ++c;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// counters (per frame): [0] n_fg (clamped) [1] overflow flag [2] n_valid clusters.  slot_over[frame]: the overflow flag
// kept with the frame's slot -- what search() of that frame reports (a batched search reports any of its frames')
__global__ __launch_bounds__(1024) void fg_scan(const int* __restrict__ blk_count, int nblk,
                                                int* __restrict__ blk_offset, int max_fg,
                                                int* __restrict__ counters,
                                                int* __restrict__ slot_n_fg, long slot_int_stride,
                                                int* __restrict__ slot_over) {
    __shared__ int wtot[16];
    __shared__ int carry;
    blk_count += (size_t)blockIdx.x * nblk;
    blk_offset += (size_t)blockIdx.x * nblk;
    counters += blockIdx.x * 4;
    slot_n_fg += blockIdx.x * slot_int_stride;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? blk_count[i] : 0;
        int incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wtot[wid] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wid; ++w) before += wtot[w];
        int total = 0;
        for (int w = 0; w < 16; ++w) total += wtot[w];
        if (i < nblk) blk_offset[i] = carry + before + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int n = carry;
        counters[1] = n > max_fg ? 1 : 0;
        counters[0] = n > max_fg ? max_fg : n;
        counters[2] = 0;
        *slot_n_fg = counters[0];
        slot_over[blockIdx.x] = counters[1];
    }
}

// locate.cpp:237-250
__global__ __launch_bounds__(256) void fg_compact(LocParams P, const float* __restrict__ diff,
                                                  size_t npx, const int* __restrict__ blk_offset,
                                                  int max_fg, int* __restrict__ fg_pixel,
                                                  float* __restrict__ fg_xyz, float* __restrict__ fg_depth,
                                                  long slot_int_stride, long slot_f_stride) {
    __shared__ int wtot[4];
    diff += (size_t)blockIdx.y * npx;
    blk_offset += (size_t)blockIdx.y * gridDim.x;
    fg_pixel += blockIdx.y * slot_int_stride;
    fg_xyz += blockIdx.y * slot_f_stride;
    fg_depth += (size_t)blockIdx.y * max_fg;
    const size_t base = (size_t)blockIdx.x * FG_PX_PER_BLOCK + (size_t)threadIdx.x * 4;
    float val[4];
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        val[j] = (base + j < npx) ? diff[base + j] : 0.0f;
        if (val[j] != 0) ++c;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wtot[wid] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wid; ++w) before += wtot[w];
    int dst = blk_offset[blockIdx.x] + before + incl - c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (val[j] != 0) {
            if (dst < max_fg) {
                const size_t p = base + j;
                const int v = (int)(p / P.wz), u = (int)(p % P.wz);
                const float uvd[3] = {(float)u, (float)v, val[j]};
                float o[3];
                camera_to_lidar(P, uvd, o);
                fg_pixel[dst] = (int)p;
                fg_depth[dst] = val[j];
                fg_xyz[dst * 3 + 0] = o[0];
                fg_xyz[dst * 3 + 1] = o[1];
                fg_xyz[dst * 3 + 2] = o[2];
            }
            ++dst;
        }
    }
}

// ---- cluster(): the whole connected-components stage in ONE workgroup ---------------------------
//
// Foreground lists are small (hundreds to a few thousand points), so six dependent grid launches
// of near-empty kernels cost more in launch gaps and global atomics than the work itself
// (cc_pairs alone measured 343 us per frame).  Here one 1024-thread workgroup does init, all-pairs
// union, flatten, size filter, ranking and id assignment with the union-find forest in LDS
// (n <= CC_LDS_MAX) -- LDS atomics instead of L2 round trips -- or, for larger n, the same code
// over the global scratch arrays.
//
// The pair tests are pruned with the ordering of the list: points are in row-major pixel order and
// two points closer than the tolerance cannot be more than R rows apart.  In the camera frame (the
// lidar frame is a rigid image of it) a point reconstructed from pixel row v at depth d has
// y = d (v / zoom - cy) / fy, so for |p - q| < tol:
//   |v_p - v_q| = zoom fy |y_p / d_p - y_q / d_q| < zoom fy tol / d_p * (1 + |y_q| / d_q),
// and |y_q| / d_q <= max(cy, H - cy) / fy.  Point i therefore only meets the candidates j < i whose
// row is within R_i = floor(row_k / d_i) + 1 of its own: a contiguous tail of the list, found by
// binary search.  The partition is exactly the all-pairs one (every pair within the tolerance is
// still tested); d_i <= 0 or a skewed intrinsic (row_k <= 0) falls back to all pairs.
constexpr int CC_THREADS = 1024;
constexpr int CC_LDS_MAX = 4096;

// LDS forest: plain (volatile) reads; global forest: L2-served atomic loads, because another
// thread's atomicCAS executes in L2 and would not refresh a line this CU holds in L1
template <bool SMALL>
__device__ __forceinline__ int cc_ld(volatile int* parent, int x) {
    if (SMALL) return parent[x];
    return __hip_atomic_load((int*)parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool SMALL>
__device__ __forceinline__ int cc_find(volatile int* parent, int x) {
    int p = cc_ld<SMALL>(parent, x);
    while (p != x) {
        x = p;
        p = cc_ld<SMALL>(parent, x);
    }
    return x;
}

template <bool SMALL>
__device__ void cc_block(int n, const float* __restrict__ xyz, float tol2, int min_size, int max_size,
                         volatile int* parent, int* csize, int* root_id, int* vroot, int* vsize,
                         const float* lxyz, int* nvalid, int* __restrict__ fg_cluster, const int* pix,
                         const float* __restrict__ depth, float row_k, int wz) {
    const int tid = threadIdx.x;
    if (SMALL) {
        for (int i = tid; i < n; i += CC_THREADS) {
            parent[i] = i;
            csize[i] = 0;
            root_id[i] = -1;
        }
    }
    if (tid == 0) *nvalid = 0;
    __syncthreads();
    // all pairs j < i; thread-strided i keeps the triangular work balanced across threads.  (Lists beyond the
    // LDS forest arrive here with the forest already built by cc_init_grid / cc_pairs_grid over the whole chip.)
    const float* src = SMALL ? lxyz : xyz;
    for (int i = tid; i < (SMALL ? n : 0); i += CC_THREADS) {
        const float ax = src[i * 3 + 0], ay = src[i * 3 + 1], az = src[i * 3 + 2];
        int my_root = i;
        int j0 = 0;
        const float d_i = depth[i];
        if (row_k > 0.f && d_i > 0.f) {
            const float rows = row_k / d_i;
            if (rows < 1.0e6f) {
                const int v_lo = pix[i] / wz - ((int)rows + 1);
                if (v_lo > 0) {  // first j with pixel >= v_lo * wz
                    const int key = v_lo * wz;
                    int lo = 0, hi = i;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (pix[mid] < key) lo = mid + 1; else hi = mid;
                    }
                    j0 = lo;
                }
            }
        }
        for (int j = j0; j < i; ++j) {
            const float dx = src[j * 3 + 0] - ax, dy = src[j * 3 + 1] - ay, dz = src[j * 3 + 2] - az;
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < tol2) {
                if (cc_ld<SMALL>(parent, j) == my_root) continue;  // already united through a common root
                int a = i, b = j;
                for (;;) {
                    a = cc_find<SMALL>(parent, a);
                    b = cc_find<SMALL>(parent, b);
                    if (a == b) break;
                    if (a < b) {
                        const int t = a;
                        a = b;
                        b = t;
                    }
                    if (atomicCAS((int*)parent + a, a, b) == a) break;
                }
                my_root = a < b ? a : b;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += CC_THREADS) atomicAdd(csize + cc_find<SMALL>(parent, i), 1);
    __syncthreads();
    for (int i = tid; i < n; i += CC_THREADS) {
        const int r = cc_find<SMALL>(parent, i);
        if (r == i) {
            const int s = csize[i];
            if (s >= min_size && s <= max_size) {
                const int k = atomicAdd(nvalid, 1);
                vroot[k] = i;
                vsize[k] = s;
            }
        }
    }
    __syncthreads();
    const int nv = *nvalid;
    for (int a = tid; a < nv; a += CC_THREADS) {
        const int ra = vroot[a], sa = vsize[a];
        int rank = 0;
        for (int b = 0; b < nv; ++b) {
            const int sb = vsize[b], rb = vroot[b];
            if (sb > sa || (sb == sa && rb < ra)) ++rank;
        }
        root_id[ra] = rank;
    }
    __syncthreads();
    for (int i = tid; i < n; i += CC_THREADS) fg_cluster[i] = root_id[cc_find<SMALL>(parent, i)];
}

// ---- lists beyond CC_LDS_MAX points (K = 20 robots of ~400 points): the pair phase over the whole chip ----------
// One workgroup cannot hide 8 k points x a few hundred candidates each (0.2 ms per frame); the union-find forest
// moves to global memory (L2 atomics: compare-and-swap hooks the larger root under the smaller, as in the LDS
// forest, so the partition and its roots = lowest member index are the same) and the points spread over the
// grid.  Both kernels return at once for a list that fits the single-workgroup path.
__global__ __launch_bounds__(256) void cc_init_grid(const int* __restrict__ counters, int* parent, int* csize, int* root_id, int max_fg) {
    counters += blockIdx.y * 4;
    parent += (size_t)blockIdx.y * max_fg, csize += (size_t)blockIdx.y * max_fg, root_id += (size_t)blockIdx.y * max_fg;
    const int n = counters[0];
    if (n <= CC_LDS_MAX) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) parent[i] = i, csize[i] = 0, root_id[i] = -1;
}

__global__ __launch_bounds__(256) void cc_pairs_grid(const int* __restrict__ counters, const float* __restrict__ xyz, float tol2,
                                                      int* parent_g, const int* __restrict__ pix,
                                                      const float* __restrict__ depth, float row_k, int wz, int max_fg,
                                                      long slot_int_stride, long slot_f_stride) {
    counters += blockIdx.y * 4;
    const int n = counters[0];
    if (n <= CC_LDS_MAX) return;
    xyz += blockIdx.y * slot_f_stride;
    pix += blockIdx.y * slot_int_stride;
    depth += (size_t)blockIdx.y * max_fg;
    volatile int* parent = parent_g + (size_t)blockIdx.y * max_fg;
    // a wave takes one point and spreads its candidates over the lanes: neighbouring points have neighbouring
    // candidate ranges, so the wave's loads are the same few cache lines
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    for (int i = wave; i < n; i += n_waves) {
        const float ax = xyz[i * 3 + 0], ay = xyz[i * 3 + 1], az = xyz[i * 3 + 2];
        int j0 = 0;
        const float d_i = depth[i];
        if (row_k > 0.f && d_i > 0.f) {
            const float rows = row_k / d_i;
            if (rows < 1.0e6f) {
                const int v_lo = pix[i] / wz - ((int)rows + 1);
                if (v_lo > 0) {
                    const int key = v_lo * wz;
                    int lo = 0, hi = i;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (pix[mid] < key) lo = mid + 1; else hi = mid;
                    }
                    j0 = lo;
                }
            }
        }
        for (int j = j0 + lane; j < i; j += 64) {
            const float dx = xyz[j * 3 + 0] - ax, dy = xyz[j * 3 + 1] - ay, dz = xyz[j * 3 + 2] - az;
            if (dx * dx + dy * dy + dz * dz < tol2) {
                int a = i, b = j;
                for (;;) {
                    a = cc_find<false>(parent, a);
                    b = cc_find<false>(parent, b);
                    if (a == b) break;
                    if (a < b) {
                        const int t = a;
                        a = b;
                        b = t;
                    }
                    if (atomicCAS((int*)parent + a, a, b) == a) break;
                }
            }
        }
    }
}

__global__ __launch_bounds__(CC_THREADS) void cc_fused(int* __restrict__ counters, const float* __restrict__ xyz,
                                                       float tol2, int min_size, int max_size,
                                                       int* g_parent, int* g_csize, int* g_root_id, int* g_vroot,
                                                       int* g_vsize, int* __restrict__ fg_cluster,
                                                       int* __restrict__ slot_n_clusters,
                                                       const int* __restrict__ fg_pixel,
                                                       const float* __restrict__ fg_depth, float row_k, int wz, int max_fg,
                                                       long slot_int_stride, long slot_f_stride) {
    extern __shared__ __attribute__((aligned(16))) int cc_lds[];
    __shared__ int nvalid;
    {
        const size_t f = blockIdx.x, fo = f * (size_t)max_fg;
        counters += f * 4;
        xyz += f * slot_f_stride;
        fg_cluster += f * slot_int_stride, slot_n_clusters += f * slot_int_stride, fg_pixel += f * slot_int_stride;
        fg_depth += fo;
        g_parent += fo, g_csize += fo, g_root_id += fo, g_vroot += fo, g_vsize += fo;
    }
    const int n = counters[0];
    if (n <= CC_LDS_MAX) {
        int* parent = cc_lds;                    // [CC_LDS_MAX]
        int* csize = cc_lds + CC_LDS_MAX;        // [CC_LDS_MAX]
        int* root_id = cc_lds + 2 * CC_LDS_MAX;  // [CC_LDS_MAX]
        int* vroot = cc_lds + 3 * CC_LDS_MAX;    // [CC_LDS_MAX]
        int* vsize = cc_lds + 4 * CC_LDS_MAX;    // [CC_LDS_MAX]
        float* lxyz = (float*)(cc_lds + 5 * CC_LDS_MAX);  // [3 * CC_LDS_MAX]
        int* lpix = cc_lds + 8 * CC_LDS_MAX;              // [CC_LDS_MAX]
        for (int i = threadIdx.x; i < 3 * n; i += CC_THREADS) lxyz[i] = xyz[i];
        for (int i = threadIdx.x; i < n; i += CC_THREADS) lpix[i] = fg_pixel[i];
        __syncthreads();
        cc_block<true>(n, xyz, tol2, min_size, max_size, parent, csize, root_id, vroot, vsize, lxyz, &nvalid, fg_cluster,
                       lpix, fg_depth, row_k, wz);
    } else {
        cc_block<false>(n, xyz, tol2, min_size, max_size, g_parent, g_csize, g_root_id, g_vroot, g_vsize, nullptr,
                        &nvalid, fg_cluster, fg_pixel, fg_depth, row_k, wz);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        counters[2] = nvalid;
        *slot_n_clusters = nvalid;
    }
}

__global__ __launch_bounds__(256) void slot_copy(const int* __restrict__ s_nfg,
                                                 const int* __restrict__ s_ncl,
                                                 const int* __restrict__ s_pix,
                                                 const float* __restrict__ s_xyz,
                                                 const int* __restrict__ s_cl,
                                                 int* __restrict__ d_nfg, int* __restrict__ d_ncl,
                                                 int* __restrict__ d_pix, float* __restrict__ d_xyz,
                                                 int* __restrict__ d_cl, const int* __restrict__ s_over,
                                                 int* __restrict__ d_over) {
    const int n = *s_nfg;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        *d_nfg = n;
        *d_ncl = *s_ncl;
        *d_over = *s_over;
    }
    if (i >= n) return;
    d_pix[i] = s_pix[i];
    d_cl[i] = s_cl[i];
    d_xyz[i * 3 + 0] = s_xyz[i * 3 + 0];
    d_xyz[i * 3 + 1] = s_xyz[i * 3 + 1];
    d_xyz[i * 3 + 2] = s_xyz[i * 3 + 2];
}

// ---- search() ------------------------------------------------------------------------------

// locate.cpp:276-311.  rects: zoomed (x,y,w,h) per robot; out: {located, x, y, z} (metres).
__global__ __launch_bounds__(256) void loc_search(LocParams P, const int* __restrict__ n_fg_p,
                                                  const int* __restrict__ n_cl_p,
                                                  const int* __restrict__ fg_pixel,
                                                  const float* __restrict__ fg_xyz,
                                                  const int* __restrict__ fg_cluster,
                                                  const int* __restrict__ rects,
                                                  float* __restrict__ out, int max_buckets,
                                                  const int* __restrict__ robot_frame, long slot_int_stride, long slot_f_stride) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    int* cnt = smem;  // [max_buckets] bucket b = cluster id b-1
    if (robot_frame) {   // a batch over kept frames: this robot's frame selects the FrameSlot (consecutive slots)
        const long f = robot_frame[blockIdx.x];
        n_fg_p += f * slot_int_stride, n_cl_p += f * slot_int_stride;
        fg_pixel += f * slot_int_stride, fg_cluster += f * slot_int_stride;
        fg_xyz += f * slot_f_stride;
    }
    __shared__ int red_cnt[4], red_key[4];
    __shared__ double red_s[4][3];
    __shared__ int win_key, win_cnt;

    const int n = *n_fg_p;
    const int nb = min(*n_cl_p + 1, max_buckets);
    const int rx = rects[blockIdx.x * 4 + 0], ry = rects[blockIdx.x * 4 + 1];
    const int rw = rects[blockIdx.x * 4 + 2], rh = rects[blockIdx.x * 4 + 3];
    for (int b = threadIdx.x; b < nb; b += 256) cnt[b] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int p = fg_pixel[i];
        const int v = p / P.wz, u = p % P.wz;
        if (u >= rx && u < rx + rw && v >= ry && v < ry + rh) atomicAdd(&cnt[fg_cluster[i] + 1], 1);
    }
    __syncthreads();
    // first max in ascending key: larger count wins, equal counts -> smaller key
    int bc = 0, bk = 0x7fffffff;
    for (int b = threadIdx.x; b < nb; b += 256) {
        const int c = cnt[b];
        if (c > bc || (c == bc && c > 0 && b < bk)) {
            bc = c;
            bk = b;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int oc = __shfl_down(bc, o), ok = __shfl_down(bk, o);
        if (oc > bc || (oc == bc && ok < bk)) {
            bc = oc;
            bk = ok;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        red_cnt[threadIdx.x >> 6] = bc;
        red_key[threadIdx.x >> 6] = bk;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = red_cnt[0], k = red_key[0];
        for (int w = 1; w < 4; ++w)
            if (red_cnt[w] > c || (red_cnt[w] == c && red_key[w] < k)) {
                c = red_cnt[w];
                k = red_key[w];
            }
        win_cnt = c;
        win_key = k;
    }
    __syncthreads();
    const int wc = win_cnt, wk = win_key;
    if (wc == 0) {
        if (threadIdx.x == 0) {
            out[blockIdx.x * 4 + 0] = 0;
            out[blockIdx.x * 4 + 1] = out[blockIdx.x * 4 + 2] = out[blockIdx.x * 4 + 3] = 0;
        }
        return;
    }
    double sx = 0, sy = 0, sz = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int p = fg_pixel[i];
        const int v = p / P.wz, u = p % P.wz;
        if (u >= rx && u < rx + rw && v >= ry && v < ry + rh && fg_cluster[i] + 1 == wk) {
            sx += (double)fg_xyz[i * 3 + 0];
            sy += (double)fg_xyz[i * 3 + 1];
            sz += (double)fg_xyz[i * 3 + 2];
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_down(sx, o);
        sy += __shfl_down(sy, o);
        sz += __shfl_down(sz, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red_s[threadIdx.x >> 6][0] = sx;
        red_s[threadIdx.x >> 6][1] = sy;
        red_s[threadIdx.x >> 6][2] = sz;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s[3];
        for (int k = 0; k < 3; ++k) s[k] = ((red_s[0][k] + red_s[1][k]) + red_s[2][k]) + red_s[3][k];
        const float loc[3] = {(float)(s[0] / (double)wc), (float)(s[1] / (double)wc),
                              (float)(s[2] / (double)wc)};
        float w[3];
        lidar_to_world(P, loc, w);
        out[blockIdx.x * 4 + 0] = 1.0f;
        for (int k = 0; k < 3; ++k) out[blockIdx.x * 4 + 1 + k] = (float)((double)w[k] * 1e-3);
    }
}

// ---- host: OpenCV-compatible inverses [OpenCV behaviour] --------------------------------------

// cv::Matx33f::inv(): closed-form adjugate in f32
bool inv3x3_f32(const float a[9], float b[9]) {
    auto A = [&](int i, int j) { return a[i * 3 + j]; };
    float d = A(0, 0) * (A(1, 1) * A(2, 2) - A(2, 1) * A(1, 2)) -
              A(0, 1) * (A(1, 0) * A(2, 2) - A(2, 0) * A(1, 2)) +
              A(0, 2) * (A(1, 0) * A(2, 1) - A(2, 0) * A(1, 1));
    if (d == 0) {
        std::memset(b, 0, 9 * sizeof(float));
        return false;
    }
    d = 1 / d;
    b[0] = (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1)) * d;
    b[1] = (A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2)) * d;
    b[2] = (A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1)) * d;
    b[3] = (A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2)) * d;
    b[4] = (A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0)) * d;
    b[5] = (A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2)) * d;
    b[6] = (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0)) * d;
    b[7] = (A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1)) * d;
    b[8] = (A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0)) * d;
    return true;
}

// cv::Matx44f::inv(): LU with partial pivoting on [A | I], f32
bool inv4x4_f32(const float a[16], float out[16]) {
    float A[4][4], B[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            A[i][j] = a[i * 4 + j];
            B[i][j] = i == j ? 1.0f : 0.0f;
        }
    for (int i = 0; i < 4; ++i) {
        int k = i;
        for (int j = i + 1; j < 4; ++j)
            if (std::fabs(A[j][i]) > std::fabs(A[k][i])) k = j;
        if (std::fabs(A[k][i]) < 1.1920929e-06f) {
            std::memset(out, 0, 16 * sizeof(float));
            return false;
        }
        if (k != i) {
            for (int j = i; j < 4; ++j) std::swap(A[i][j], A[k][j]);
            for (int j = 0; j < 4; ++j) std::swap(B[i][j], B[k][j]);
        }
        const float d = -1 / A[i][i];
        for (int j = i + 1; j < 4; ++j) {
            const float alpha = A[j][i] * d;
            for (int q = i + 1; q < 4; ++q) A[j][q] += alpha * A[i][q];
            for (int q = 0; q < 4; ++q) B[j][q] += alpha * B[i][q];
        }
    }
    for (int i = 3; i >= 0; --i)
        for (int j = 0; j < 4; ++j) {
            float s = B[i][j];
            for (int k = i + 1; k < 4; ++k) s -= A[i][k] * B[k][j];
            B[i][j] = s / A[i][i];
        }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[i * 4 + j] = B[i][j];
    return true;
}

// ---- host: Locator ----------------------------------------------------------------------------

static inline int cv_round(float v) { return (int)lrintf(v); }  // saturate_cast<int>(float)

// locate.cpp:112-146
Locator::Locator(const rmr_locator_cfg& cfg) : cfg_(cfg), ctx_(device_ctx(cfg.device)) {
    if (cfg.image_width <= 0 || cfg.image_height <= 0)
        fail(RMR_ERR_INVALID_ARGUMENT, "Locator: image size must be positive");
    if (cfg.queue_size < 1 || cfg.queue_size > MAX_QUEUE)
        fail(RMR_ERR_INVALID_ARGUMENT, "Locator: queue_size must be in 1..%d", MAX_QUEUE);
    if (cfg_.max_points <= 0) cfg_.max_points = 262144;
    if (cfg_.max_foreground <= 0) cfg_.max_foreground = 32768;
    if (cfg_.max_frames <= 0) cfg_.max_frames = 1;
    prm_.zoom = cfg.zoom_factor;
    prm_.wz = (int)((float)cfg.image_width * cfg.zoom_factor);
    prm_.hz = (int)((float)cfg.image_height * cfg.zoom_factor);
    if (prm_.wz <= 0 || prm_.hz <= 0) fail(RMR_ERR_INVALID_ARGUMENT, "Locator: zoomed image is empty");
    std::memcpy(prm_.K, cfg.intrinsic, sizeof(prm_.K));
    std::memcpy(prm_.L2C, cfg.lidar_to_camera, sizeof(prm_.L2C));
    inv3x3_f32(prm_.K, prm_.Kinv);
    float c2l[16], c2w[16];
    inv4x4_f32(prm_.L2C, c2l);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) prm_.R[i * 3 + j] = c2l[i * 4 + j];
        prm_.t[i] = c2l[i * 4 + 3];
    }
    inv4x4_f32(cfg.world_to_camera, c2w);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0;
            for (int k = 0; k < 4; ++k) s += c2w[i * 4 + k] * prm_.L2C[k * 4 + j];
            prm_.L2W[i * 4 + j] = s;
        }
    prm_.min_diff = cfg.min_depth_diff;
    prm_.max_diff = cfg.max_depth_diff;
    prm_.max_distance = cfg.max_distance;
    prm_.tol2 = cfg.cluster_tolerance * cfg.cluster_tolerance;
    prm_.min_cluster = cfg.min_cluster_size;
    prm_.max_cluster = cfg.max_cluster_size;

    npx_ = (size_t)prm_.wz * prm_.hz;
    const int mf = cfg_.max_foreground;
    max_clusters_ = mf / (cfg.min_cluster_size > 1 ? cfg.min_cluster_size : 1);
    if ((size_t)(max_clusters_ + 1) * sizeof(int) > 150 * 1024)
        fail(RMR_ERR_INVALID_ARGUMENT,
             "Locator: max_foreground/min_cluster_size = %d buckets exceed the LDS histogram",
             max_clusters_);

    RMR_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    key_.alloc(npx_);
    bg_.alloc(npx_);
    diff_.alloc(npx_);
    ring_.alloc(npx_ * cfg.queue_size);
    RMR_HIP(hipMemsetAsync(key_.p, 0, npx_ * sizeof(unsigned long long), stream_));
    RMR_HIP(hipMemsetAsync(bg_.p, 0, npx_ * sizeof(float), stream_));  // Q12: zero-initialised
    RMR_HIP(hipMemsetAsync(diff_.p, 0, npx_ * sizeof(float), stream_));
    RMR_HIP(hipMemsetAsync(ring_.p, 0, npx_ * cfg.queue_size * sizeof(float), stream_));
    cloud_.alloc((size_t)cfg_.max_points * 4);
    cloud_pin_.alloc((size_t)cfg_.max_points * 4);

    const int nblk = (int)((npx_ + FG_PX_PER_BLOCK - 1) / FG_PX_PER_BLOCK);
    // cluster() scratch, one set per frame of a batch (update_cluster_batch runs max_frames frames in one launch each)
    const size_t bf = (size_t)cfg_.max_frames;
    blk_count_.alloc(bf * nblk);
    blk_offset_.alloc(bf * nblk);
    fg_depth_.alloc(bf * mf);
    // the global union-find forest is only touched by lists beyond CC_LDS_MAX points
    const size_t forest = mf > CC_LDS_MAX ? bf * mf : (size_t)mf;
    parent_.alloc(forest);
    csize_.alloc(forest);
    vroot_.alloc(forest);
    vsize_.alloc(forest);
    root_id_.alloc(forest);
    counters_.alloc(4 * bf);
    RMR_HIP(hipMemsetAsync(counters_.p, 0, 4 * bf * sizeof(int), stream_));
    slot_over_.alloc(1 + bf);   // [0] the current frame, [1 + f] kept frame f
    RMR_HIP(hipMemsetAsync(slot_over_.p, 0, (1 + bf) * sizeof(int), stream_));

    const int nslots = 1 + cfg_.max_frames;
    slot_ints_.alloc((size_t)nslots * (2 + 2 * (size_t)mf));
    slot_floats_.alloc((size_t)nslots * 3 * (size_t)mf);
    RMR_HIP(hipMemsetAsync(slot_ints_.p, 0, slot_ints_.n * sizeof(int), stream_));
    for (int s = 0; s < nslots; ++s) {
        FrameSlot f;
        int* base = slot_ints_.p + (size_t)s * (2 + 2 * (size_t)mf);
        f.n_fg = base;
        f.n_clusters = base + 1;
        f.fg_pixel = base + 2;
        f.fg_cluster = base + 2 + mf;
        f.fg_xyz = slot_floats_.p + (size_t)s * 3 * mf;
        slots_.push_back(f);
    }
    rects_dev_.alloc(4 * 256);
    loc_dev_.alloc(4 * 256);
    rects_pin_.alloc(4 * 256);
    loc_pin_.alloc(4 * 256);
    RMR_HIP(hipStreamSynchronize(stream_));
}

Locator::~Locator() {
    if (stream_) {
        (void)hipStreamSynchronize(stream_);
        (void)hipStreamDestroy(stream_);
    }
}

// locate.cpp:158-220
void Locator::update(const float* xyz, int n, int stride_bytes, int mem) { update_into(xyz, n, stride_bytes, mem, diff_.p); }

void Locator::update_into(const float* xyz, int n, int stride_bytes, int mem, float* diff_out) {
    ctx_.use();
    if (!xyz || n <= 0) {
        // locate.cpp:160-171: depth and diff cleared, nothing queued
        RMR_HIP(hipMemsetAsync(diff_out, 0, npx_ * sizeof(float), stream_));
        return;
    }
    if (stride_bytes < 12 || (stride_bytes & 3))
        fail(RMR_ERR_INVALID_ARGUMENT, "Locator::update: stride_bytes must be a multiple of 4, >= 12");
    if (n > cfg_.max_points) fail(RMR_ERR_CAPACITY, "Locator::update: %d points exceed max_points=%d", n, cfg_.max_points);
    const char* dev_xyz;
    if (mem == RMR_MEM_DEVICE) {
        dev_xyz = (const char*)xyz;
    } else {
        const size_t bytes = (size_t)n * stride_bytes;
        if (bytes > cloud_.n * sizeof(float)) {
            cloud_.alloc(bytes / 4 + 4);
            cloud_pin_.alloc(bytes / 4 + 4);
        }
        // the pinned staging buffer is reused: wait for the previous frame's copy
        RMR_HIP(hipStreamSynchronize(stream_));
        std::memcpy(cloud_pin_.p, xyz, bytes);
        RMR_HIP(hipMemcpyAsync(cloud_.p, cloud_pin_.p, bytes, hipMemcpyHostToDevice, stream_));
        dev_xyz = (const char*)cloud_.p;
    }
    {
        ProfScope ps(ctx_.prof, stream_, "loc_scatter", 0, (double)n * 16);
        loc_scatter<<<(n + 255) / 256, 256, 0, stream_>>>(prm_, dev_xyz, n, stride_bytes, key_.p, bg_.p);
        RMR_HIP(hipGetLastError());
    }
    // locate.cpp:195-198: push_back, pop_front when over queue_size
    const int Q = cfg_.queue_size;
    int newest;
    if (ring_len_ < Q) {
        newest = (ring_head_ + ring_len_) % Q;
        ++ring_len_;
    } else {
        newest = ring_head_;
        ring_head_ = (ring_head_ + 1) % Q;
    }
    RingOrder order{};
    order.n = ring_len_;
    order.newest = newest;
    for (int i = 0; i < ring_len_; ++i) order.slot[i] = (ring_head_ + i) % Q;
    {
        ProfScope ps(ctx_.prof, stream_, "loc_diff", 0, (double)npx_ * (8 + 4 * (ring_len_ + 2)));
        loc_diff<<<(unsigned)((npx_ + 255) / 256), 256, 0, stream_>>>(prm_, order, key_.p, bg_.p, ring_.p, diff_out, npx_);
        RMR_HIP(hipGetLastError());
    }
}

// locate.cpp:231-264
// smallest singular value of a 3x3 matrix (closed-form eigenvalues of M^T M): lidar-frame distances
// are at least this factor times camera-frame distances
static double min_singular3(const float* m) {
    double a[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            a[i][j] = 0;
            for (int k = 0; k < 3; ++k) a[i][j] += (double)m[k * 3 + i] * m[k * 3 + j];
        }
    const double p1 = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double q = (a[0][0] + a[1][1] + a[2][2]) / 3.0;
    const double p2 = (a[0][0] - q) * (a[0][0] - q) + (a[1][1] - q) * (a[1][1] - q) + (a[2][2] - q) * (a[2][2] - q) + 2 * p1;
    if (p2 < 1e-30) return std::sqrt(std::max(q, 0.0));  // a multiple of the identity
    const double p = std::sqrt(p2 / 6.0);
    double b[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) b[i][j] = (a[i][j] - (i == j ? q : 0.0)) / p;
    const double det = b[0][0] * (b[1][1] * b[2][2] - b[1][2] * b[2][1]) - b[0][1] * (b[1][0] * b[2][2] - b[1][2] * b[2][0]) +
                       b[0][2] * (b[1][0] * b[2][1] - b[1][1] * b[2][0]);
    const double r = std::min(1.0, std::max(-1.0, det / 2.0));
    const double phi = std::acos(r) / 3.0;
    const double e_min = q + 2.0 * p * std::cos(phi + 2.0943951023931953);
    return std::sqrt(std::max(e_min, 0.0)) * 0.999;
}

void Locator::cluster() { cluster_frames(diff_.p, 1, 0); }

// The frames' foreground images lie npx_ floats apart from `diff`; their products go to slots first_slot, first_slot + 1, ...
// Every kernel takes the frame from its block index, so a batch costs the launches of one frame.
void Locator::cluster_frames(const float* diff, int n_frames, int first_slot) {
    ctx_.use();
    const int mf = cfg_.max_foreground;
    const int nblk = (int)((npx_ + FG_PX_PER_BLOCK - 1) / FG_PX_PER_BLOCK);
    const int gfg = (mf + 255) / 256;
    const unsigned F = (unsigned)n_frames;
    const long sis = 2 + 2 * (long)mf, sfs = 3 * (long)mf;   // ints / floats from one FrameSlot to the next
    FrameSlot& cur = slots_[first_slot];
    ProfScope ps(ctx_.prof, stream_, "loc_cluster", 0, (double)npx_ * 8 * n_frames);
    fg_count<<<dim3(nblk, F), 256, 0, stream_>>>(diff, npx_, blk_count_.p);
    fg_scan<<<F, 1024, 0, stream_>>>(blk_count_.p, nblk, blk_offset_.p, mf, counters_.p, cur.n_fg, sis, slot_over_.p + first_slot);
    fg_compact<<<dim3(nblk, F), 256, 0, stream_>>>(prm_, diff, npx_, blk_offset_.p, mf, cur.fg_pixel, cur.fg_xyz, fg_depth_.p, sis, sfs);
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)cc_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * CC_LDS_MAX * 4);
    });
    // rows two points closer than the tolerance can be apart, times the depth (see cc_block);
    // only for a pinhole intrinsic whose image row depends on y alone
    float row_k = 0.f;
    const double s_min = min_singular3(prm_.R);  // 1 for a calibration (rigid) camera-to-lidar rotation
    if (prm_.K[3] == 0.f && prm_.K[4] > 0.f && s_min > 1e-3) {
        const float fy = prm_.K[4], cy = prm_.K[5], h_full = (float)prm_.hz / prm_.zoom;
        const float t_max = std::max(std::fabs(cy), std::fabs(h_full - cy)) / fy;
        row_k = (float)(prm_.zoom * fy * std::sqrt(prm_.tol2) / s_min * (1.f + t_max) * 1.001);
    }
    if (mf > CC_LDS_MAX) {   // lists that can outgrow the single-workgroup forest: pair phase over the chip
        cc_init_grid<<<dim3(gfg, F), 256, 0, stream_>>>(counters_.p, parent_.p, csize_.p, root_id_.p, mf);
        // a frame's pair phase gets the whole chip when it is alone, its share of it in a batch (the kernel strides)
        const int pw = std::max(8, std::min(gfg * 4, 4 * ctx_.num_cus) / n_frames);
        cc_pairs_grid<<<dim3(pw, F), 256, 0, stream_>>>(counters_.p, cur.fg_xyz, prm_.tol2, parent_.p, cur.fg_pixel, fg_depth_.p, row_k,
                                                       prm_.wz, mf, sis, sfs);
    }
    cc_fused<<<F, CC_THREADS, 9 * CC_LDS_MAX * sizeof(int), stream_>>>(
        counters_.p, cur.fg_xyz, prm_.tol2, prm_.min_cluster, prm_.max_cluster, parent_.p, csize_.p, root_id_.p,
        vroot_.p, vsize_.p, cur.fg_cluster, cur.n_clusters, cur.fg_pixel, fg_depth_.p, row_k, prm_.wz, mf, sis, sfs);
    RMR_HIP(hipGetLastError());
}

// update + cluster + keep(f) of n_frames consecutive frames of this stream.  The updates stay one after another (the
// background image and the depth ring are the stream's history), each leaving its foreground image in its own slice
// of diff_batch_; then ONE cluster pass over all frames writes the kept slots 0 .. n_frames-1 directly.  Per frame
// the same kernels see the same inputs as update(); cluster(); keep(f), so the results are the same bits -- but a
// 64-frame batch is 2 x 64 + 8 launches instead of 7 x 64, and the single-workgroup connected-components stage runs
// as one 64-workgroup launch instead of 64 launches that each hold a CU for ~0.1 ms beside the detector's kernels.
void Locator::update_cluster_batch(const float* const* clouds, const int* n_points, int stride_bytes, int mem, int n_frames) {
    ctx_.use();
    if (n_frames <= 0) return;
    if (n_frames > cfg_.max_frames)
        fail(RMR_ERR_INVALID_ARGUMENT, "Locator: a batch of %d frames exceeds max_frames=%d", n_frames, cfg_.max_frames);
    if (diff_batch_.n < (size_t)n_frames * npx_) diff_batch_.alloc((size_t)cfg_.max_frames * npx_);
    static const bool per_frame = std::getenv("RMR_LOC_BATCH_UPDATE") && std::atoi(std::getenv("RMR_LOC_BATCH_UPDATE")) == 0;
    if (per_frame || n_frames == 1 || cfg_.queue_size > 8) {
        for (int f = 0; f < n_frames; ++f) update_into(clouds[f], n_points[f], stride_bytes, mem, diff_batch_.p + (size_t)f * npx_);
    } else {
        update_batch(clouds, n_points, stride_bytes, mem, n_frames);
    }
    cluster_frames(diff_batch_.p, n_frames, 1);
    // "the current frame" (read_image, foreground(), search(slot -1)) is the batch's last one
    RMR_HIP(hipMemcpyAsync(diff_.p, diff_batch_.p + (size_t)(n_frames - 1) * npx_, npx_ * sizeof(float), hipMemcpyDeviceToDevice, stream_));
    const FrameSlot& s = slots_[n_frames];
    const FrameSlot& d = slots_[0];
    slot_copy<<<(cfg_.max_foreground + 255) / 256, 256, 0, stream_>>>(
        s.n_fg, s.n_clusters, s.fg_pixel, s.fg_xyz, s.fg_cluster, d.n_fg, d.n_clusters, d.fg_pixel, d.fg_xyz, d.fg_cluster,
        slot_over_.p + n_frames, slot_over_.p);
    RMR_HIP(hipGetLastError());
}

// The update stage of a batch as two launches (loc_scatter_batch, loc_walk_batch) instead of two per frame.
void Locator::update_batch(const float* const* clouds, const int* n_points, int stride_bytes, int mem, int n_frames) {
    if (stride_bytes < 12 || (stride_bytes & 3))
        fail(RMR_ERR_INVALID_ARGUMENT, "Locator::update: stride_bytes must be a multiple of 4, >= 12");
    const size_t F = (size_t)cfg_.max_frames;
    if (key_batch_.n < F * npx_) {
        key_batch_.alloc(F * npx_);
        fmax_batch_.alloc(F * npx_);
        RMR_HIP(hipMemsetAsync(key_batch_.p, 0, key_batch_.n * sizeof(unsigned long long), stream_));
        RMR_HIP(hipMemsetAsync(fmax_batch_.p, 0, fmax_batch_.n * sizeof(int), stream_));
        table_dev_.alloc(F * sizeof(CloudTable));
        table_pin_.alloc(F * sizeof(CloudTable));
    }
    // the pinned table (and the pinned cloud block) are reused from batch to batch: the previous batch's copies must be done
    RMR_HIP(hipStreamSynchronize(stream_));
    CloudTable* tab = (CloudTable*)table_pin_.p;
    int max_n = 0;
    size_t host_bytes = 0;
    for (int f = 0; f < n_frames; ++f) {
        const int n = clouds[f] && n_points[f] > 0 ? n_points[f] : 0;
        if (n > cfg_.max_points) fail(RMR_ERR_CAPACITY, "Locator::update: %d points exceed max_points=%d", n, cfg_.max_points);
        max_n = std::max(max_n, n);
        tab[f].n = n;
        tab[f].pad = 0;
        tab[f].xyz = n ? (const char*)clouds[f] : nullptr;
        if (n && mem != RMR_MEM_DEVICE) host_bytes += ((size_t)n * stride_bytes + 255) & ~(size_t)255;
    }
    if (host_bytes) {   // host clouds: one pinned block, one copy for the batch
        if (host_bytes > cloud_.n * sizeof(float)) {
            cloud_.alloc(host_bytes / 4 + 64);
            cloud_pin_.alloc(host_bytes / 4 + 64);
        }
        size_t off = 0;
        for (int f = 0; f < n_frames; ++f) {
            if (!tab[f].n) continue;
            const size_t bytes = (size_t)tab[f].n * stride_bytes;
            std::memcpy((char*)cloud_pin_.p + off, clouds[f], bytes);
            tab[f].xyz = (const char*)cloud_.p + off;
            off += (bytes + 255) & ~(size_t)255;
        }
        RMR_HIP(hipMemcpyAsync(cloud_.p, cloud_pin_.p, off, hipMemcpyHostToDevice, stream_));
    }
    RMR_HIP(hipMemcpyAsync(table_dev_.p, table_pin_.p, (size_t)n_frames * sizeof(CloudTable), hipMemcpyHostToDevice, stream_));
    const CloudTable* dtab = (const CloudTable*)table_dev_.p;
    if (max_n > 0) {
        ProfScope ps(ctx_.prof, stream_, "loc_scatter", 0, 0);
        loc_scatter_batch<<<dim3((max_n + 255) / 256, n_frames), 256, 0, stream_>>>(prm_, dtab, stride_bytes, key_batch_.p, fmax_batch_.p, npx_);
        RMR_HIP(hipGetLastError());
    }
    {
        ProfScope ps(ctx_.prof, stream_, "loc_diff", 0, (double)npx_ * n_frames * 16);
        const unsigned g = (unsigned)((npx_ + 255) / 256);
#define RMR_WALK(QQ) loc_walk_batch<QQ><<<g, 256, 0, stream_>>>(prm_, dtab, n_frames, key_batch_.p, fmax_batch_.p, bg_.p, ring_.p, ring_head_, ring_len_, diff_batch_.p, npx_)
        switch (cfg_.queue_size) {
            case 1: RMR_WALK(1); break;
            case 2: RMR_WALK(2); break;
            case 3: RMR_WALK(3); break;
            case 4: RMR_WALK(4); break;
            case 5: RMR_WALK(5); break;
            case 6: RMR_WALK(6); break;
            case 7: RMR_WALK(7); break;
            default: RMR_WALK(8); break;
        }
#undef RMR_WALK
        RMR_HIP(hipGetLastError());
    }
    // the ring now lies oldest first from slot 0
    int valid = 0;
    for (int f = 0; f < n_frames; ++f) valid += tab[f].n > 0;
    ring_len_ = std::min(cfg_.queue_size, ring_len_ + valid);
    ring_head_ = 0;
}

void Locator::keep(int frame) {
    ctx_.use();
    if (frame < 0 || frame >= cfg_.max_frames)
        fail(RMR_ERR_INVALID_ARGUMENT, "Locator::keep: frame %d out of range (max_frames=%d)", frame, cfg_.max_frames);
    const FrameSlot& s = slots_[0];
    const FrameSlot& d = slots_[1 + frame];
    slot_copy<<<(cfg_.max_foreground + 255) / 256, 256, 0, stream_>>>(
        s.n_fg, s.n_clusters, s.fg_pixel, s.fg_xyz, s.fg_cluster, d.n_fg, d.n_clusters, d.fg_pixel, d.fg_xyz, d.fg_cluster,
        slot_over_.p, slot_over_.p + 1 + frame);
    RMR_HIP(hipGetLastError());
}

// locate.cpp:337-350 (Q19)
void Locator::zoom(const int rect[4], int out[4]) const {
    const float z = prm_.zoom;
    const float center_x = (float)rect[0] * z + (float)rect[2] * z * 0.5f;
    const float center_y = (float)rect[1] * z + (float)rect[3] * z * 0.5f;
    const int ret_width = (int)((float)rect[2] * z);
    const int ret_height = (int)((float)rect[3] * z);
    const int ret_x = (int)(center_x - (float)ret_width * 0.5f);
    const int ret_y = (int)(center_y - (float)ret_height * 0.5f);
    const int x1 = std::max(ret_x, 0), y1 = std::max(ret_y, 0);
    const int x2 = std::min(ret_x + ret_width, prm_.wz), y2 = std::min(ret_y + ret_height, prm_.hz);
    if (x2 - x1 <= 0 || y2 - y1 <= 0) {
        out[0] = out[1] = out[2] = out[3] = 0;
        return;
    }
    out[0] = x1;
    out[1] = y1;
    out[2] = x2 - x1;
    out[3] = y2 - y1;
}

// locate.cpp:323-326 over locate.cpp:276-311
void Locator::search(rmr_robot* robots, int n, int slot) {
    ctx_.use();
    if (n <= 0) return;
    if (slot < -1 || slot >= cfg_.max_frames)
        fail(RMR_ERR_INVALID_ARGUMENT, "Locator::search: frame %d out of range", slot);
    const FrameSlot& f = slots_[slot + 1];
    rects_pin_.ensure((size_t)4 * n);
    loc_pin_.ensure((size_t)4 * n);
    rects_dev_.ensure((size_t)4 * n);
    loc_dev_.ensure((size_t)4 * n);
    for (int i = 0; i < n; ++i) {
        // Robot::rect(): Rect2f -> Rect rounds half to even (robot.h:111)
        const int ri[4] = {cv_round(robots[i].rect[0]), cv_round(robots[i].rect[1]),
                           cv_round(robots[i].rect[2]), cv_round(robots[i].rect[3])};
        zoom(ri, rects_pin_.p + 4 * i);
    }
    RMR_HIP(hipMemcpyAsync(rects_dev_.p, rects_pin_.p, sizeof(int) * 4 * n, hipMemcpyHostToDevice, stream_));
    {
        ProfScope ps(ctx_.prof, stream_, "loc_search", 0, 0);
        const int nbuckets = max_clusters_ + 1;
        loc_search<<<n, 256, nbuckets * sizeof(int), stream_>>>(prm_, f.n_fg, f.n_clusters, f.fg_pixel, f.fg_xyz,
                                                                f.fg_cluster, rects_dev_.p, loc_dev_.p, nbuckets, nullptr, 0, 0);
        RMR_HIP(hipGetLastError());
    }
    RMR_HIP(hipMemcpyAsync(loc_pin_.p, loc_dev_.p, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, stream_));
    int flags[2] = {0, 0};
    RMR_HIP(hipMemcpyAsync(flags, slot_over_.p + slot + 1, sizeof(int), hipMemcpyDeviceToHost, stream_));  // this frame's flag
    RMR_HIP(hipStreamSynchronize(stream_));
    for (int i = 0; i < n; ++i) {
        const float* o = loc_pin_.p + 4 * i;
        if (o[0] != 0) {
            robots[i].has_location = 1;
            robots[i].location[0] = o[1];
            robots[i].location[1] = o[2];
            robots[i].location[2] = o[3];
        }
    }
    if (flags[0]) fail(RMR_ERR_CAPACITY, "Locator: foreground exceeded max_foreground=%d points", cfg_.max_foreground);
}

void Locator::search_batch(rmr_robot* robots, const int* counts, int n_frames, int cap) {
    search_batch_begin(robots, counts, n_frames, cap);
    search_batch_end(robots, counts, n_frames, cap);
}

static int checked_total(const rmr_robot* robots, const int* counts, int n_frames, int cap, int max_frames) {
    if (!robots || !counts || cap <= 0 || n_frames > max_frames)
        fail(RMR_ERR_INVALID_ARGUMENT, "Locator::search_batch: bad arguments (frames %d of %d kept slots)", n_frames, max_frames);
    int total = 0;
    for (int f = 0; f < n_frames; ++f) {
        if (counts[f] < 0 || counts[f] > cap) fail(RMR_ERR_INVALID_ARGUMENT, "Locator::search_batch: counts[%d] = %d", f, counts[f]);
        total += counts[f];
    }
    return total;
}

void Locator::search_batch_begin(const rmr_robot* robots, const int* counts, int n_frames, int cap) {
    ctx_.use();
    if (n_frames <= 0) return;
    const int total = checked_total(robots, counts, n_frames, cap, cfg_.max_frames);
    if (total == 0) return;
    // staging: 4 ints of zoomed rect per robot, then one int per robot: the kept frame it belongs to
    rects_pin_.ensure((size_t)5 * total);
    loc_pin_.ensure((size_t)4 * total);
    rects_dev_.ensure((size_t)5 * total);
    loc_dev_.ensure((size_t)4 * total);
    search_flags_.ensure((size_t)n_frames);
    int at = 0;
    for (int f = 0; f < n_frames; ++f)
        for (int i = 0; i < counts[f]; ++i, ++at) {
            const rmr_robot& r = robots[(size_t)f * cap + i];
            const int ri[4] = {cv_round(r.rect[0]), cv_round(r.rect[1]), cv_round(r.rect[2]), cv_round(r.rect[3])};
            zoom(ri, rects_pin_.p + 4 * at);
            rects_pin_.p[4 * total + at] = f;
        }
    RMR_HIP(hipMemcpyAsync(rects_dev_.p, rects_pin_.p, sizeof(int) * 5 * total, hipMemcpyHostToDevice, stream_));
    const int nbuckets = max_clusters_ + 1;
    {
        // ONE launch over the robots of all frames (a workgroup per robot, as in search())
        const FrameSlot& s = slots_[1];
        const long mf = cfg_.max_foreground;
        ProfScope ps(ctx_.prof, stream_, "loc_search", 0, 0);
        loc_search<<<total, 256, nbuckets * sizeof(int), stream_>>>(prm_, s.n_fg, s.n_clusters, s.fg_pixel, s.fg_xyz, s.fg_cluster,
                                                                    rects_dev_.p, loc_dev_.p, nbuckets, rects_dev_.p + 4 * total,
                                                                    2 + 2 * mf, 3 * mf);
        RMR_HIP(hipGetLastError());
    }
    RMR_HIP(hipMemcpyAsync(loc_pin_.p, loc_dev_.p, sizeof(float) * 4 * total, hipMemcpyDeviceToHost, stream_));
    RMR_HIP(hipMemcpyAsync(search_flags_.p, slot_over_.p + 1, sizeof(int) * n_frames, hipMemcpyDeviceToHost, stream_));
}

void Locator::search_batch_end(rmr_robot* robots, const int* counts, int n_frames, int cap) {
    ctx_.use();
    if (n_frames <= 0) return;
    const int total = checked_total(robots, counts, n_frames, cap, cfg_.max_frames);
    if (total == 0) return;
    RMR_HIP(hipStreamSynchronize(stream_));
    int at = 0;
    for (int f = 0; f < n_frames; ++f)
        for (int i = 0; i < counts[f]; ++i, ++at) {
            const float* o = loc_pin_.p + 4 * at;
            if (o[0] != 0) {
                rmr_robot& r = robots[(size_t)f * cap + i];
                r.has_location = 1;
                r.location[0] = o[1], r.location[1] = o[2], r.location[2] = o[3];
            }
        }
    for (int f = 0; f < n_frames; ++f)
        if (search_flags_.p[f])
            fail(RMR_ERR_CAPACITY, "Locator: foreground exceeded max_foreground=%d points (frame %d of the batch)", cfg_.max_foreground, f);
}

float* Locator::image_ptr(int which) {
    switch (which) {
        case RMR_LOC_DEPTH: {
            const int Q = cfg_.queue_size;
            const int newest = ring_len_ ? (ring_head_ + ring_len_ - 1) % Q : 0;
            return ring_.p + (size_t)newest * npx_;
        }
        case RMR_LOC_BACKGROUND: return bg_.p;
        case RMR_LOC_DIFF: return diff_.p;
    }
    fail(RMR_ERR_INVALID_ARGUMENT, "Locator: unknown image id %d", which);
}

void Locator::read_image(int which, float* host_out) {
    ctx_.use();
    RMR_HIP(hipMemcpyAsync(host_out, image_ptr(which), npx_ * sizeof(float), hipMemcpyDeviceToHost, stream_));
    RMR_HIP(hipStreamSynchronize(stream_));
}

void Locator::write_image(int which, const float* host_in) {
    ctx_.use();
    RMR_HIP(hipMemcpyAsync(image_ptr(which), host_in, npx_ * sizeof(float), hipMemcpyHostToDevice, stream_));
    RMR_HIP(hipStreamSynchronize(stream_));
}

// ---- snapshot of the temporal state -------------------------------------------------------------
// header (8 ints: magic, version, zoomed width, height, queue_size, valid ring slots, 0, 0), the
// background image, then the valid depth images oldest first.
namespace {
constexpr int kStateMagic = 0x4c524d52;  // "RMRL"
constexpr int kStateVersion = 1;
}  // namespace

size_t Locator::state_bytes() const { return 8 * sizeof(int) + (size_t)(1 + cfg_.queue_size) * npx_ * sizeof(float); }

void Locator::save_state(void* host_out, size_t cap) {
    if (!host_out || cap < state_bytes()) fail(RMR_ERR_CAPACITY, "Locator::save_state: buffer of %zu bytes, %zu needed", cap, state_bytes());
    ctx_.use();
    int* hdr = (int*)host_out;
    const int h[8] = {kStateMagic, kStateVersion, prm_.wz, prm_.hz, cfg_.queue_size, ring_len_, 0, 0};
    std::copy(h, h + 8, hdr);
    float* img = (float*)(hdr + 8);
    RMR_HIP(hipMemcpyAsync(img, bg_.p, npx_ * sizeof(float), hipMemcpyDeviceToHost, stream_));
    for (int i = 0; i < cfg_.queue_size; ++i) {
        float* dst = img + (size_t)(1 + i) * npx_;
        if (i < ring_len_) {
            const int slot = (ring_head_ + i) % cfg_.queue_size;
            RMR_HIP(hipMemcpyAsync(dst, ring_.p + (size_t)slot * npx_, npx_ * sizeof(float), hipMemcpyDeviceToHost, stream_));
        } else {
            std::fill(dst, dst + npx_, 0.f);
        }
    }
    RMR_HIP(hipStreamSynchronize(stream_));
}

void Locator::load_state(const void* host_in, size_t bytes) {
    if (!host_in || bytes < 8 * sizeof(int)) fail(RMR_ERR_INVALID_ARGUMENT, "Locator::load_state: truncated snapshot");
    const int* hdr = (const int*)host_in;
    if (hdr[0] != kStateMagic || hdr[1] != kStateVersion) fail(RMR_ERR_INVALID_ARGUMENT, "Locator::load_state: not a locator snapshot");
    if (hdr[2] != prm_.wz || hdr[3] != prm_.hz || hdr[4] != cfg_.queue_size)
        fail(RMR_ERR_INVALID_ARGUMENT, "Locator::load_state: snapshot is %d x %d with %d depth images, this locator %d x %d with %d",
             hdr[2], hdr[3], hdr[4], prm_.wz, prm_.hz, cfg_.queue_size);
    if (bytes < state_bytes() || hdr[5] < 0 || hdr[5] > cfg_.queue_size) fail(RMR_ERR_INVALID_ARGUMENT, "Locator::load_state: truncated snapshot");
    ctx_.use();
    const float* img = (const float*)(hdr + 8);
    RMR_HIP(hipMemcpyAsync(bg_.p, img, npx_ * sizeof(float), hipMemcpyHostToDevice, stream_));
    RMR_HIP(hipMemcpyAsync(ring_.p, img + npx_, (size_t)cfg_.queue_size * npx_ * sizeof(float), hipMemcpyHostToDevice, stream_));
    RMR_HIP(hipStreamSynchronize(stream_));
    ring_head_ = 0;
    ring_len_ = hdr[5];
}

void Locator::transform(int which, const float in[3], float out[3]) const {
    switch (which) {
        case RMR_XF_LIDAR_TO_WORLD: lidar_to_world(prm_, in, out); return;
        case RMR_XF_CAMERA_TO_LIDAR: camera_to_lidar(prm_, in, out); return;
        case RMR_XF_LIDAR_TO_CAMERA: lidar_to_camera(prm_, in, out); return;
    }
    fail(RMR_ERR_INVALID_ARGUMENT, "Locator: unknown transform id %d", which);
}

void Locator::foreground(float* xyz, int* pixel, int* cluster, int cap, int* n) {
    ctx_.use();
    const FrameSlot& f = slots_[0];
    int hdr[2] = {0, 0};
    RMR_HIP(hipMemcpyAsync(hdr, f.n_fg, sizeof(int) * 2, hipMemcpyDeviceToHost, stream_));
    RMR_HIP(hipStreamSynchronize(stream_));
    const int m = std::min(hdr[0], cap);
    *n = hdr[0];
    if (m > 0) {
        if (xyz) RMR_HIP(hipMemcpyAsync(xyz, f.fg_xyz, sizeof(float) * 3 * m, hipMemcpyDeviceToHost, stream_));
        if (pixel) RMR_HIP(hipMemcpyAsync(pixel, f.fg_pixel, sizeof(int) * m, hipMemcpyDeviceToHost, stream_));
        if (cluster) RMR_HIP(hipMemcpyAsync(cluster, f.fg_cluster, sizeof(int) * m, hipMemcpyDeviceToHost, stream_));
        RMR_HIP(hipStreamSynchronize(stream_));
    }
}

int Locator::num_clusters() {
    ctx_.use();
    int v = 0;
    RMR_HIP(hipMemcpyAsync(&v, slots_[0].n_clusters, sizeof(int), hipMemcpyDeviceToHost, stream_));
    RMR_HIP(hipStreamSynchronize(stream_));
    return v;
}

}  // namespace rmr
