// reduce.hip -- range reductions and order statistics for gfx950.
//
//   minmax_t / minmax_c  one streaming pass (16-B loads), wave64 shuffle reduction, LDS across the
//                        4 waves of a workgroup, ONE pair of global atomics per workgroup on plain
//                        float storage (sign-split integer min/max).  Replaces the
//                        transpose+flatten+torch.min/max of TorchMinMaxObserver.observe
//                        (ppq/quantization/observer/range.py:86-98).
//   (quantile_t lives in quantile.hip)
//   isotone_t            replaces Isotone_T (sort.cu:61-73): top-2 / bottom-2 reduction.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.hpp"

namespace ppqhip {

// ------------------------------------------------------------------------------------ min / max
__device__ __forceinline__ void block_minmax_commit(float mn, float mx, float* gmin, float* gmax, float* lds) {
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    if (lane == 0) { lds[wid] = mn; lds[8 + wid] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; w++) { mn = fminf(mn, lds[w]); mx = fmaxf(mx, lds[8 + w]); }
        if (mn <= mx) {   // false only when this workgroup saw no (non-NaN) element
            atomic_min_f32(gmin, mn);
            atomic_max_f32(gmax, mx);
        }
    }
}

// partial == nullptr: commit with one pair of global atomics per workgroup (few workgroups);
// otherwise write this workgroup's (min, max) to partial[2*blockIdx.x ..] for minmax_finish_kernel:
// same-address device atomics serialise at ~12 ns each, which would dominate a 2048-workgroup launch.
template <int U, bool NT>
__global__ __launch_bounds__(kBlock) void minmax_t_kernel(const float* __restrict__ x, uint32_t n, int vec_ok,
                                                          float* __restrict__ minmax, float* __restrict__ partial,
                                                          int accumulate) {
    __shared__ float lds[16];
    float mn = INFINITY, mx = -INFINITY;
    stream_elems<U, NT>(x, n, vec_ok != 0, [&](float a) { mn = fminf(mn, a); mx = fmaxf(mx, a); });
    if (partial == nullptr) { block_minmax_commit(mn, mx, &minmax[0], &minmax[1], lds); return; }
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { lds[wid] = mn; lds[8 + wid] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / kWave; w++) { mn = fminf(mn, lds[w]); mx = fmaxf(mx, lds[8 + w]); }
        if (accumulate) {   // persistent per-workgroup slot: race-free, stream-ordered read-modify-write
            mn = fminf(mn, partial[2 * blockIdx.x]);
            mx = fmaxf(mx, partial[2 * blockIdx.x + 1]);
        }
        partial[2 * blockIdx.x] = mn;
        partial[2 * blockIdx.x + 1] = mx;
    }
}

// many tensors, one launch -- the persistent design of hist_persistent_kernel (hist.hip): the work is the
// concatenated list of tiles (kMMBlock * kMMU float4) of all jobs, split evenly over a chip-sized grid;
// a workgroup walks its contiguous tile range with two ping-pong register tiles and folds its running
// (min, max) into ITS OWN slot of every job it touches (plain stream-ordered read-modify-write).
constexpr int kMinMaxMultiMax = 96;              // jobs per launch (2.3 KB of kernel arguments)
#ifndef PPQHIP_MM_BLOCK
#define PPQHIP_MM_BLOCK 256
#endif
#ifndef PPQHIP_MM_STREAM
#define PPQHIP_MM_STREAM 1                 // single tensors beyond 16 MB: the ping-pong form of minmax_small_kernel
#endif
#ifndef PPQHIP_MM_WGPC
#define PPQHIP_MM_WGPC 8
#endif
#ifndef PPQHIP_MM_U
#define PPQHIP_MM_U 2
#endif
constexpr int kMMBlock = PPQHIP_MM_BLOCK, kMMU = PPQHIP_MM_U;
constexpr int kMMGrid = kNumCU * PPQHIP_MM_WGPC;                 // <= ppqhip_minmax_slots()
constexpr uint32_t kMMTileVec = (uint32_t)kMMBlock * kMMU, kMMTileElems = kMMTileVec * 4;
static_assert(kMMGrid <= kNumCU * 8, "one slot per workgroup");
struct MinMaxJob {
    const float* x;
    float* slots;
    uint32_t n;
    uint32_t first_tile;
};
struct MinMaxJobs {
    MinMaxJob job[kMinMaxMultiMax];
    uint32_t count;
    uint32_t total_tiles;
};
__host__ __device__ inline uint32_t mm_job_tiles(uint32_t n, bool vec_ok) {
    if (!vec_ok) return (n + kMMTileElems - 1) / kMMTileElems;
    const uint32_t full = (n >> 2) / kMMTileVec;
    return full + (n > full * kMMTileElems ? 1u : 0u);
}

template <bool NT>
__global__ __launch_bounds__(kMMBlock) void minmax_persistent_kernel(const MinMaxJobs jobs) {
    __shared__ float lds[32];
    const uint32_t G = gridDim.x, g = blockIdx.x;
    uint32_t t, t_end;
    even_split(jobs.total_tiles, G, g, t, t_end);
    if (t >= t_end) return;
    uint32_t lo = 0, hi = jobs.count;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs.job[mid].first_tile <= t) lo = mid; else hi = mid;
    }
    for (uint32_t j = lo; t < t_end; j++) {
        const MinMaxJob& job = jobs.job[j];
        const uint32_t j_end = (j + 1 < jobs.count) ? jobs.job[j + 1].first_tile : jobs.total_tiles;
        uint32_t k = t - job.first_tile;
        const uint32_t k1 = min(t_end, j_end) - job.first_tile;
        t = min(t_end, j_end);
        const float* __restrict__ x = job.x;
        const uint32_t n = job.n;
        const bool vec_ok = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
        const uint32_t full = vec_ok ? (n >> 2) / kMMTileVec : 0u;
        const uint32_t kf = min(k1, full);
        float mn = INFINITY, mx = -INFINITY;
        if (k < kf) {
            const float4* xv = reinterpret_cast<const float4*>(x) + threadIdx.x;
            float4 bufa[kMMU], bufb[kMMU];
            auto fetch = [&](float4 (&buf)[kMMU], uint32_t tile) {
                const float4* p = xv + (size_t)tile * kMMTileVec;
#pragma unroll
                for (int u = 0; u < kMMU; u++) buf[u] = load4<NT>(p + u * kMMBlock);
            };
            auto consume = [&](const float4 (&buf)[kMMU]) {
#pragma unroll
                for (int u = 0; u < kMMU; u++) {
                    mn = fminf(fminf(mn, buf[u].x), fminf(buf[u].y, fminf(buf[u].z, buf[u].w)));
                    mx = fmaxf(fmaxf(mx, buf[u].x), fmaxf(buf[u].y, fmaxf(buf[u].z, buf[u].w)));
                }
            };
            fetch(bufa, k);
            for (;;) {
                fetch(bufb, min(k + 1, kf - 1));
                consume(bufa);
                if (++k >= kf) break;
                fetch(bufa, min(k + 1, kf - 1));
                consume(bufb);
                if (++k >= kf) break;
            }
        }
        for (; k < k1; k++) {             // ragged tail tile / unaligned tensor: masked 4-B loads
            const uint32_t e0 = k * kMMTileElems + threadIdx.x;
#pragma unroll 4
            for (int r = 0; r < 4 * kMMU; r++) {
                const uint32_t i = e0 + r * kMMBlock;
                if (i < n) { const float a = x[i]; mn = fminf(mn, a); mx = fmaxf(mx, a); }
            }
        }
        mn = wave_min(mn);
        mx = wave_max(mx);
        const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) { lds[wid] = mn; lds[16 + wid] = mx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kMMBlock / kWave; w++) { mn = fminf(mn, lds[w]); mx = fmaxf(mx, lds[16 + w]); }
            job.slots[2 * g] = fminf(mn, job.slots[2 * g]);
            job.slots[2 * g + 1] = fmaxf(mx, job.slots[2 * g + 1]);
        }
        __syncthreads();
    }
}

// The single-tensor launch for LATENCY-bound sizes (see hist_small_kernel, hist.hip: direct arguments instead of the 2.3 KB job
// table and its two dependent scalar-load rounds, every load of the share in flight before anything else).  The workgroup's slot
// is read FIRST, so the read-modify-write at the end is no dependent load -> store chain.  K: rows of kMMBlock float4 per share.
// PING (mid-size and large single tensors, round 6): two register tiles of K rows ping-pong -- the persistent kernel's streaming loop
// without its job table.  A small gain here, unlike the histogram's (the persistent min/max kernel was within 4-10 % of the read
// floor already): interleaved A/B Bx4 6.28 -> 5.92 us, Bx16 19.9 -> 19.3, Bx32 32.4 -> 31.9 (0.805 of 8 TB/s), Bx8 equal at the
// 10th percentile (10.3 us), profiles/r06_minmax_stream_ab.txt.
template <int K, bool PING = false, bool NT = false>
__global__ __launch_bounds__(kMMBlock) void minmax_small_kernel(const float* __restrict__ x, uint32_t n, float* __restrict__ slots) {
    __shared__ float lds[32];
    const uint32_t G = gridDim.x, g = blockIdx.x;
    float s_mn = INFINITY, s_mx = -INFINITY;
    if (threadIdx.x == 0) { s_mn = slots[2 * g]; s_mx = slots[2 * g + 1]; }
    const uint32_t nvec = n >> 2, full_rows = nvec / kMMBlock;
    uint32_t r0, r1;
    even_split(full_rows, G, g, r0, r1);
    const float4* xv = reinterpret_cast<const float4*>(x) + threadIdx.x;
    float mn = INFINITY, mx = -INFINITY;
    auto fetch = [&](float4 (&buf)[K], uint32_t r) {
#pragma unroll
        for (int k = 0; k < K; k++) buf[k] = load4<NT>(xv + (size_t)min(r + (uint32_t)k, r1 - 1) * kMMBlock);      // clamped: duplicates are harmless
    };
    auto fold = [&](const float4 (&buf)[K]) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            mn = fminf(fminf(mn, buf[k].x), fminf(buf[k].y, fminf(buf[k].z, buf[k].w)));
            mx = fmaxf(fmaxf(mx, buf[k].x), fmaxf(buf[k].y, fmaxf(buf[k].z, buf[k].w)));
        }
    };
    if (PING) {
        if (r0 < r1) {
            float4 bufa[K], bufb[K];
            uint32_t r = r0;
            fetch(bufa, r);
            for (;;) {
                fetch(bufb, r + K);
                fold(bufa);
                r += K;
                if (r >= r1) break;
                fetch(bufa, r + K);
                fold(bufb);
                r += K;
                if (r >= r1) break;
            }
        }
    } else {
        for (uint32_t r = r0; r < r1; r += K) {
            float4 buf[K];
            fetch(buf, r);
            fold(buf);
        }
    }
    if (g == G - 1) {                                                  // the ragged rest: < kMMBlock float4 + n % 4 elements
        for (uint32_t i = full_rows * kMMBlock * 4 + threadIdx.x; i < n; i += kMMBlock) { const float a = x[i]; mn = fminf(mn, a); mx = fmaxf(mx, a); }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { lds[wid] = mn; lds[16 + wid] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kMMBlock / kWave; w++) { mn = fminf(mn, lds[w]); mx = fmaxf(mx, lds[16 + w]); }
        slots[2 * g] = fminf(mn, s_mn);
        slots[2 * g + 1] = fmaxf(mx, s_mx);
    }
}

__global__ __launch_bounds__(kBlock) void minmax_finish_kernel(const float* __restrict__ partial, uint32_t count,
                                                               float* __restrict__ minmax) {
    __shared__ float lds[16];
    float mn = INFINITY, mx = -INFINITY;
    for (uint32_t i = threadIdx.x; i < count; i += kBlock) {
        mn = fminf(mn, partial[2 * i]);
        mx = fmaxf(mx, partial[2 * i + 1]);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { lds[wid] = mn; lds[8 + wid] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / kWave; w++) { mn = fminf(mn, lds[w]); mx = fmaxf(mx, lds[8 + w]); }
        minmax[0] = fminf(minmax[0], mn);   // stream-ordered read-modify-write (accumulate semantics)
        minmax[1] = fmaxf(minmax[1], mx);
    }
}

// rows of `epc` contiguous elements, ONE WAVE per work item, no workgroup barrier anywhere: item i = (channel c = i % C,
// row group g = i / C, chunk) -> the wave reduces `K` rows of its channel (rows g*K .. of stride C) or one 8192-element
// chunk of one row, with U 16-B loads in flight per lane, folds across its 64 lanes with shuffles and commits ONE pair of
// atomics.  (Round 2's workgroup-per-row kernel spent a barrier, an LDS round trip and two atomics on every 12.5 KB row
// of a [32, 512, 56, 56] activation and kept one load per lane in flight: 0.68 of the roofline, this one 0.73.  Sweep in
// profiles/r03_minmax_c_variants.txt: U = 4 / 8 / 16, 8 .. 64 waves per CU, predicated instead of clamped loads -- all within
// 35.1 .. 37.2 us; shipped: U = 16 (a 56 x 56 row is one trip) and >= 16 waves per CU.)
#ifndef PPQHIP_MMC_U
#define PPQHIP_MMC_U 16
#endif
#ifndef PPQHIP_MMC_WPC
#define PPQHIP_MMC_WPC 16            // waves per CU a launch should at least have before a wave takes several rows
#endif
constexpr int kMMCU = PPQHIP_MMC_U;
constexpr uint32_t kMMCChunk = 8192;            // elements per chunk of a long row
__global__ __launch_bounds__(kBlock) void minmax_c_wave_kernel(const float* __restrict__ x, uint32_t epc, int vec_ok, uint32_t C,
                                                               uint32_t outer, uint32_t K, uint32_t chunks, uint32_t items,
                                                               float* __restrict__ mins, float* __restrict__ maxs) {
    const uint32_t item = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (item >= items) return;                                  // wave-uniform
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t rc = item / chunks, chunk = item - rc * chunks;      // chunks == 1 unless K == 1
    const uint32_t g = rc / C, c = rc - g * C;
    const uint32_t n0 = g * K, n1 = min(n0 + K, outer);
    const uint32_t lo = chunk * kMMCChunk, hi = min(lo + kMMCChunk, epc);
    float mn = INFINITY, mx = -INFINITY;
    for (uint32_t n = n0; n < n1; n++) {
        const float* xr = x + ((size_t)n * C + c) * epc;
        if (vec_ok) {   // epc % 4 == 0, base 16-B aligned
            const float4* xv = reinterpret_cast<const float4*>(xr);
            const uint32_t v1 = hi >> 2;
            for (uint32_t v0 = (lo >> 2) + lane; v0 < v1; v0 += 64 * kMMCU) {
                float4 a[kMMCU];
#pragma unroll
                for (int u = 0; u < kMMCU; u++) a[u] = xv[min(v0 + 64 * u, v1 - 1)];       // clamped: loads stay unconditional
#pragma unroll
                for (int u = 0; u < kMMCU; u++) {
                    mn = fminf(fminf(mn, a[u].x), fminf(a[u].y, fminf(a[u].z, a[u].w)));
                    mx = fmaxf(fmaxf(mx, a[u].x), fmaxf(a[u].y, fmaxf(a[u].z, a[u].w)));
                }
            }
        } else {
            for (uint32_t j = lo + lane; j < hi; j += 64) { const float a = xr[j]; mn = fminf(mn, a); mx = fmaxf(mx, a); }
        }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane == 0 && mn <= mx) { atomic_min_f32(&mins[c], mn); atomic_max_f32(&maxs[c], mx); }
}

// short rows (channel-last, [N,C,1,1] ...): per-element LDS (or global) atomics by channel
__global__ __launch_bounds__(kBlock) void minmax_c_generic_kernel(const float* __restrict__ x, uint32_t n,
                                                                  FastDiv elem_per_channel, FastDiv num_channel,
                                                                  int use_lds, float* __restrict__ mins,
                                                                  float* __restrict__ maxs) {
    extern __shared__ float mm[];   // [C] mins, [C] maxs
    const uint32_t C = num_channel.d;
    if (use_lds) {
        for (uint32_t c = threadIdx.x; c < C; c += kBlock) { mm[c] = INFINITY; mm[C + c] = -INFINITY; }
        __syncthreads();
    }
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float a = x[i];
        if (a != a) continue;
        const uint32_t row = fdiv(i, elem_per_channel);
        const uint32_t c = row - fdiv(row, num_channel) * C;
        if (use_lds) { atomic_min_f32(&mm[c], a); atomic_max_f32(&mm[C + c], a); }
        else { atomic_min_f32(&mins[c], a); atomic_max_f32(&maxs[c], a); }
    }
    if (use_lds) {
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < C; c += kBlock) {
            if (mm[c] <= mm[C + c]) { atomic_min_f32(&mins[c], mm[c]); atomic_max_f32(&maxs[c], mm[C + c]); }
        }
    }
}

// ------------------------------------------------------------------------------------ channel sums
// sums[c] (+)= sum over every element of channel c, accumulated in DOUBLE and reduced in a fixed
// order (deterministic, no atomics): the per-channel DC term of BiasCorrectionPass
// (ppq/quantization/optim/training.py:438-448, torch.mean over all dims but the channel one).
//   long rows  (epc >= 64): grid = (C, S); workgroup (c, s) walks rows n = s, s + S, ... of channel c
//                           with 16-B loads, block-reduces and stores partial[s][c]; the finish
//                           kernel adds the S partials of a channel in index order.
//   short rows (Gemm [N, C], channel-last ...): one thread per channel, strided rows.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__global__ __launch_bounds__(kBlock) void channel_sum_row_kernel(const float* __restrict__ x, uint32_t rows_per_channel,
                                                                 uint32_t C, uint32_t epc, int vec_ok,
                                                                 double* __restrict__ partial) {
    __shared__ double lds[kBlock / kWave];
    const uint32_t c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    double acc = 0.0;
    for (uint32_t n = s; n < rows_per_channel; n += S) {
        const float* xr = x + ((size_t)n * C + c) * epc;
        if (vec_ok) {
            const float4* xv = reinterpret_cast<const float4*>(xr);
            const uint32_t v1 = epc >> 2;
            for (uint32_t v = threadIdx.x; v < v1; v += 4 * kBlock) {          // 4 loads in flight per lane (a 56 x 56 row: one trip)
                float4 a[4];
#pragma unroll
                for (int u = 0; u < 4; u++) a[u] = xv[min(v + u * kBlock, v1 - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (v + u * kBlock < v1) acc += ((double)a[u].x + (double)a[u].y) + ((double)a[u].z + (double)a[u].w);
            }
        } else {
            for (uint32_t j = threadIdx.x; j < epc; j += kBlock) acc += (double)xr[j];
        }
    }
    acc = wave_sum_f64(acc);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) lds[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = lds[0];
        for (int w = 1; w < kBlock / kWave; w++) t += lds[w];
        partial[(size_t)s * C + c] = t;
    }
}

// One workgroup per channel, no partials, no second launch: for tensors with enough channels to fill the chip on their own
// (C >= 2 per CU).  Two rows of the channel in flight per trip (8 16-B loads per lane); the rows are added in index order, the
// lanes' sums in a fixed tree: deterministic like the two-launch form, whose dependent finish launch costs 4.7 us whatever it reads.
__global__ __launch_bounds__(kBlock) void channel_sum_single_kernel(const float* __restrict__ x, uint32_t rows_per_channel, uint32_t C,
                                                                    uint32_t epc, double* __restrict__ sums) {
    __shared__ double lds[kBlock / kWave];
    const uint32_t c = blockIdx.x;
    const uint32_t v1 = epc >> 2;                                      // (vec_ok is a precondition of this kernel)
    double acc = 0.0;
    const double seed = threadIdx.x == 0 ? sums[c] : 0.0;             // in flight with the data
    for (uint32_t n = 0; n < rows_per_channel; n += 2) {
        const float4* x0 = reinterpret_cast<const float4*>(x + ((size_t)n * C + c) * epc);
        const float4* x1 = reinterpret_cast<const float4*>(x + ((size_t)min(n + 1, rows_per_channel - 1) * C + c) * epc);
        const bool two = n + 1 < rows_per_channel;
        for (uint32_t v = threadIdx.x; v < v1; v += 4 * kBlock) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { a[u] = x0[min(v + u * kBlock, v1 - 1)]; b[u] = x1[min(v + u * kBlock, v1 - 1)]; }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (v + u * kBlock < v1) acc += ((double)a[u].x + (double)a[u].y) + ((double)a[u].z + (double)a[u].w);
            if (two) {
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (v + u * kBlock < v1) acc += ((double)b[u].x + (double)b[u].y) + ((double)b[u].z + (double)b[u].w);
            }
        }
    }
    acc = wave_sum_f64(acc);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) lds[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = lds[0];
        for (int w = 1; w < kBlock / kWave; w++) t += lds[w];
        sums[c] = seed + t;
    }
}

__global__ __launch_bounds__(kBlock) void channel_sum_finish_kernel(const double* __restrict__ partial, uint32_t S,
                                                                    uint32_t C, double* __restrict__ sums) {
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    double t = 0.0;
    for (uint32_t s = 0; s < S; s++) t += partial[(size_t)s * C + c];
    sums[c] += t;
}

__global__ __launch_bounds__(kBlock) void channel_sum_generic_kernel(const float* __restrict__ x, uint32_t outer,
                                                                     uint32_t C, uint32_t epc,
                                                                     double* __restrict__ sums) {
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    double t = 0.0;
    for (uint32_t n = 0; n < outer; n++) {
        const float* xr = x + ((size_t)n * C + c) * epc;
        for (uint32_t j = 0; j < epc; j++) t += (double)xr[j];
    }
    sums[c] += t;
}

// ------------------------------------------------------------------------------------ isotone
__device__ __forceinline__ bool aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct Top2 { float a1, a2, b1, b2; };   // a1 >= a2 largest two, b1 <= b2 smallest two (with multiplicity)

__device__ __forceinline__ void top2_push(Top2& t, float v) {
    if (v > t.a1) { t.a2 = t.a1; t.a1 = v; } else if (v > t.a2) t.a2 = v;
    if (v < t.b1) { t.b2 = t.b1; t.b1 = v; } else if (v < t.b2) t.b2 = v;
}
__device__ __forceinline__ void top2_merge(Top2& t, const Top2& o) {
    // largest two of {t.a1, t.a2, o.a1, o.a2}
    const float hi = fmaxf(t.a1, o.a1);
    const float second = fmaxf(fminf(t.a1, o.a1), fmaxf(t.a2, o.a2));
    t.a1 = hi; t.a2 = second;
    const float lo = fminf(t.b1, o.b1);
    const float second_lo = fminf(fmaxf(t.b1, o.b1), fminf(t.b2, o.b2));
    t.b1 = lo; t.b2 = second_lo;
}
__device__ __forceinline__ Top2 top2_wave(Top2 t) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        Top2 o;
        o.a1 = __shfl_xor(t.a1, m, 64); o.a2 = __shfl_xor(t.a2, m, 64);
        o.b1 = __shfl_xor(t.b1, m, 64); o.b2 = __shfl_xor(t.b2, m, 64);
        top2_merge(t, o);
    }
    return t;
}

// stage 0: x -> partial[gridDim.x]; stage 1 (one workgroup): partial -> dest
__global__ __launch_bounds__(kBlock) void isotone_kernel(const float* __restrict__ x, uint32_t n,
                                                         const Top2* __restrict__ partial_in, uint32_t n_partial,
                                                         Top2* __restrict__ partial_out, float* __restrict__ dest,
                                                         uint32_t n_total) {
    __shared__ Top2 lds[kBlock / kWave];
    Top2 t{-INFINITY, -INFINITY, INFINITY, INFINITY};
    if (partial_in == nullptr) {
        stream_elems<4>(x, n, aligned16_dev(x), [&](float v) { top2_push(t, v); });
    } else {
        for (uint32_t i = threadIdx.x; i < n_partial; i += kBlock) top2_merge(t, partial_in[i]);
    }
    t = top2_wave(t);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) lds[wid] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / kWave; w++) top2_merge(t, lds[w]);
        if (dest == nullptr) partial_out[blockIdx.x] = t;
        else if (n_total == 1) { dest[0] = dest[1] = dest[2] = dest[3] = t.a1; }
        else { dest[0] = t.a1; dest[1] = t.a2; dest[2] = t.b1; dest[3] = t.b2; }
    }
}

constexpr int kIsotoneBlocks = 1024;
static_assert(kIsotoneBlocks * sizeof(Top2) <= 65536, "the partials must fit ppqhip_quantile_workspace_bytes (quantile.hip: >= 74 KB)");

static int validate(int64_t n, const char* what) {
    if (n <= 0) { set_error("%s: tensor is empty", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (n > 0x7fffffffLL) { set_error("%s: too many elements", what); return PPQHIP_ERR_INVALID_VALUE; }
    return PPQHIP_OK;
}

// ---- per-channel min / max of MANY tensors, one launch ----------------------------------------------
// ParameterQuantizePass (optim/parameters.py:156-215) observes every weight of the graph once: 54 tensors of 0.01 .. 9 MB
// for ResNet-50, each a ~7 us launch in the per-tensor path (profiles/r03_bench_kernel_stats.csv: 270 minmax_c launches).
// One launch here: the job table travels BY VALUE in the kernel arguments (64 jobs x 52 B = 3.3 KB of the 4 KB limit, like
// HistJobs / LsqMultiArgs): no host-to-device copy, so the launch is legal inside a HIP-graph capture and does not synchronise
// the host (round 4 uploaded a pageable table per call -- ADVICE r4).  ONE WAVE per item as in minmax_c_wave_kernel:
// item -> (job, row, chunk of <= 8192 elements of that row).
// A job whose channels each consist of exactly one item (outer == 1, one chunk: every convolution / Gemm weight with
// channel axis 0) may be marked `fresh`: its wave STORES min / max, so the caller need not seed the buffers with +-inf
// (three tiny launches per weight otherwise); all other jobs fold into seeded buffers with the float atomics.
// min / max are order independent: results are bit-identical to ppqhip_minmax_c whatever the geometry.
constexpr int kMMCMultiMax = 64;
struct MMCJob {                              // 48 B
    const float* x;
    float* mins;
    float* maxs;
    uint32_t C, epc, outer, chunks;
    uint32_t vec_ok, fresh;
};
struct MMCMultiArgs {
    MMCJob jobs[kMMCMultiMax];
    uint32_t first_item[kMMCMultiMax];
    uint32_t count;
    uint32_t items;
};
static_assert(sizeof(MMCMultiArgs) <= 4096, "kernel arguments are limited to 4 KB");

__global__ __launch_bounds__(kBlock) void minmax_c_multi_kernel(const MMCMultiArgs args) {
    const uint32_t item = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (item >= args.items) return;                                  // wave-uniform
    const uint32_t lane = threadIdx.x & 63;
    uint32_t lo = 0, hi = args.count;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (args.first_item[mid] <= item) lo = mid; else hi = mid;
    }
    const MMCJob& j = args.jobs[__builtin_amdgcn_readfirstlane(lo)];  // wave-uniform index into the kernel arguments: scalar loads
    const uint32_t local = item - args.first_item[lo];
    const uint32_t row = local / j.chunks, chunk = local - row * j.chunks;
    const uint32_t c = row % j.C;
    const uint32_t e0 = chunk * kMMCChunk, e1 = min(e0 + kMMCChunk, j.epc);
    const float* xr = j.x + (size_t)row * j.epc;
    float mn = INFINITY, mx = -INFINITY;
    if (j.vec_ok) {   // epc % 4 == 0, base 16-B aligned
        const float4* xv = reinterpret_cast<const float4*>(xr);
        const uint32_t v1 = e1 >> 2;
        constexpr int U = 8;
        for (uint32_t v0 = (e0 >> 2) + lane; v0 < v1; v0 += 64 * U) {
            float4 a[U];
#pragma unroll
            for (int u = 0; u < U; u++) a[u] = xv[min(v0 + 64 * u, v1 - 1)];       // clamped: loads stay unconditional
#pragma unroll
            for (int u = 0; u < U; u++) {
                mn = fminf(fminf(mn, a[u].x), fminf(a[u].y, fminf(a[u].z, a[u].w)));
                mx = fmaxf(fmaxf(mx, a[u].x), fmaxf(a[u].y, fmaxf(a[u].z, a[u].w)));
            }
        }
    } else {
        for (uint32_t e = e0 + lane; e < e1; e += 64) { const float a = xr[e]; mn = fminf(mn, a); mx = fmaxf(mx, a); }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane == 0) {
        if (j.fresh) { j.mins[c] = mn; j.maxs[c] = mx; }                // the only item of its channel (host-checked)
        else if (mn <= mx) { atomic_min_f32(&j.mins[c], mn); atomic_max_f32(&j.maxs[c], mx); }
    }
}

}  // namespace ppqhip

using namespace ppqhip;

extern "C" {

int64_t ppqhip_minmax_workspace_bytes(int64_t n) { (void)n; return (int64_t)sizeof(float) * 2 * kNumCU * 8; }

static int minmax_t_impl(const float* x, int64_t n, float* minmax, void* workspace, float* slots, void* stream) {
    if (int st = validate(n, "minmax_t")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_MINMAX_T, 4.0 * (double)n, s);
    // 8 workgroups per CU, each streaming one contiguous chunk; streaming (nontemporal) loads once the
    // tensor cannot be cache resident (sweep on MI355X, 205 MB: 5.7 TB/s vs 5.1 with plain loads)
    const bool nt = n >= (48ll << 20);
    const int grid = stream_grid(n, kBlock * 4 * 4, num_cu() * 8);
    float* partial = slots;
    const int accumulate = slots != nullptr;
    if (!slots && grid > 32) {
        partial = workspace ? (float*)workspace : (float*)scratch(s, sizeof(float) * 2 * (size_t)grid);
        if (partial == nullptr) return PPQHIP_ERR_HIP;
    }
    if (nt)
        hipLaunchKernelGGL((minmax_t_kernel<4, true>), dim3(grid), dim3(kBlock), 0, s, x, (uint32_t)n,
                           aligned16(x) ? 1 : 0, minmax, partial, accumulate);
    else
        hipLaunchKernelGGL((minmax_t_kernel<4, false>), dim3(grid), dim3(kBlock), 0, s, x, (uint32_t)n,
                           aligned16(x) ? 1 : 0, minmax, partial, accumulate);
    if (partial && !accumulate)
        hipLaunchKernelGGL(minmax_finish_kernel, dim3(1), dim3(kBlock), 0, s, (const float*)partial, (uint32_t)grid,
                           minmax);
    return finish_launch("minmax_t");
}

int ppqhip_minmax_t(const float* x, int64_t n, float* minmax, void* workspace, void* stream) {
    return minmax_t_impl(x, n, minmax, workspace, nullptr, stream);
}

/* persistent-slot variant: slots is float[ppqhip_minmax_slots()][2], seeded with {+inf, -inf} */
int64_t ppqhip_minmax_slots(void) { return (int64_t)kNumCU * 8; }

static void launch_minmax_persistent(const MinMaxJobs& args, int64_t elems, hipStream_t s) {
    uint32_t grid = args.total_tiles / 2;                       // at least two tiles per workgroup
    if (grid < 1) grid = 1;
    const uint32_t mm_cap = (uint32_t)(num_cu() * PPQHIP_MM_WGPC);      // <= kMMGrid (one slot per workgroup)
    if (grid > mm_cap) grid = mm_cap;
    // streaming (nontemporal) loads once the data cannot be cache resident
    if (elems >= (48ll << 20)) hipLaunchKernelGGL((minmax_persistent_kernel<true>), dim3(grid), dim3(kMMBlock), 0, s, args);
    else hipLaunchKernelGGL((minmax_persistent_kernel<false>), dim3(grid), dim3(kMMBlock), 0, s, args);
}

int ppqhip_minmax_t_slots(const float* x, int64_t n, float* slots, void* stream) {
    if (slots == nullptr) { set_error("minmax_t_slots: slots is null"); return PPQHIP_ERR_INVALID_VALUE; }
    if (int st = validate(n, "minmax_t_slots")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_MINMAX_T, 4.0 * (double)n, s);
    if (n <= (4ll << 20) && aligned16(x)) {                       // up to 16 MB: the lean kernel, two rows (8 KB) per workgroup
        const uint32_t full_rows = (uint32_t)((n >> 2) / kMMBlock);
        uint32_t grid = (full_rows + 1) / 2;
        const uint32_t cap = (uint32_t)(num_cu() * PPQHIP_MM_WGPC);     // one slot per workgroup
        if (grid > cap) grid = cap;
        if (grid < 1) grid = 1;
        const uint32_t share = (full_rows + grid - 1) / grid;
        if (share <= 2) hipLaunchKernelGGL((minmax_small_kernel<2>), dim3(grid), dim3(kMMBlock), 0, s, x, (uint32_t)n, slots);
        else if (share <= 4) hipLaunchKernelGGL((minmax_small_kernel<4>), dim3(grid), dim3(kMMBlock), 0, s, x, (uint32_t)n, slots);
        else hipLaunchKernelGGL((minmax_small_kernel<8>), dim3(grid), dim3(kMMBlock), 0, s, x, (uint32_t)n, slots);
        return finish_launch("minmax_t_slots");
    }
#if PPQHIP_MM_STREAM
    if (aligned16(x)) {                                           // beyond 16 MB: the same kernel with two ping-pong tiles of two rows
        const uint32_t full_rows = (uint32_t)((n >> 2) / kMMBlock);
        uint32_t grid = full_rows / 8;                            // >= two tile pairs per workgroup
        const uint32_t cap = (uint32_t)(num_cu() * PPQHIP_MM_WGPC);
        if (grid > cap) grid = cap;
        if (grid < 1) grid = 1;
        if (n >= (48ll << 20)) hipLaunchKernelGGL((minmax_small_kernel<2, true, true>), dim3(grid), dim3(kMMBlock), 0, s, x, (uint32_t)n, slots);
        else hipLaunchKernelGGL((minmax_small_kernel<2, true, false>), dim3(grid), dim3(kMMBlock), 0, s, x, (uint32_t)n, slots);
        return finish_launch("minmax_t_slots");
    }
#endif
    MinMaxJobs args;
    args.count = 1;
    args.job[0].x = x; args.job[0].slots = slots; args.job[0].n = (uint32_t)n; args.job[0].first_tile = 0;
    args.total_tiles = mm_job_tiles((uint32_t)n, aligned16(x));
    launch_minmax_persistent(args, n, s);
    return finish_launch("minmax_t_slots");
}

int ppqhip_minmax_t_slots_multi(const ppqhip_minmax_job* jobs, int num_jobs, void* stream) {
    if (num_jobs <= 0) return PPQHIP_OK;
    if (jobs == nullptr) { set_error("minmax_t_slots_multi: jobs is null"); return PPQHIP_ERR_INVALID_VALUE; }
    double bytes = 0.0;
    for (int k = 0; k < num_jobs; k++) {
        if (int st = validate(jobs[k].n, "minmax_t_slots_multi")) return st;
        if (jobs[k].x == nullptr || jobs[k].slots == nullptr) {
            set_error("minmax_t_slots_multi: job %d has a null pointer", k); return PPQHIP_ERR_INVALID_VALUE;
        }
        bytes += 4.0 * (double)jobs[k].n;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_MINMAX_T, bytes, s);
    for (int base = 0; base < num_jobs; base += kMinMaxMultiMax) {
        MinMaxJobs args;
        args.count = (uint32_t)((num_jobs - base) < kMinMaxMultiMax ? (num_jobs - base) : kMinMaxMultiMax);
        uint32_t tiles = 0;
        int64_t elems = 0;
        for (uint32_t k = 0; k < args.count; k++) {
            const ppqhip_minmax_job& src = jobs[base + k];
            args.job[k].x = src.x; args.job[k].slots = src.slots; args.job[k].n = (uint32_t)src.n;
            args.job[k].first_tile = tiles;
            tiles += mm_job_tiles((uint32_t)src.n, aligned16(src.x));
            elems += src.n;
        }
        args.total_tiles = tiles;
        launch_minmax_persistent(args, elems, s);
    }
    return finish_launch("minmax_t_slots_multi");
}

int ppqhip_minmax_slots_finish(const float* slots, float* minmax, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(minmax_finish_kernel, dim3(1), dim3(kBlock), 0, s, slots, (uint32_t)ppqhip_minmax_slots(), minmax);
    return finish_launch("minmax_slots_finish");
}

int ppqhip_minmax_c(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel, float* mins,
                    float* maxs, void* stream) {
    if (int st = validate(n, "minmax_c")) return st;
    if (num_channel <= 0 || elem_per_channel <= 0 || n % (num_channel * elem_per_channel) != 0) {
        set_error("minmax_c: bad channel geometry"); return PPQHIP_ERR_INVALID_VALUE;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_MINMAX_C, 4.0 * (double)n, s);
    const FastDiv nc = make_fastdiv((uint32_t)num_channel);
    if (elem_per_channel >= 64) {
        const int64_t rows = n / elem_per_channel;
        const int vec_ok = (aligned16(x) && elem_per_channel % 4 == 0) ? 1 : 0;
        const uint32_t C = (uint32_t)num_channel, outer = (uint32_t)(rows / num_channel);
        const uint32_t chunks = (uint32_t)((elem_per_channel + kMMCChunk - 1) / kMMCChunk);
        // short rows: K rows of a channel per wave while that leaves the chip >= 32 waves per CU
        uint32_t K = 1;
        if (chunks == 1) {
            K = (uint32_t)(rows / (num_cu() * PPQHIP_MMC_WPC));
            const uint32_t k_bytes = (uint32_t)(65536 / (elem_per_channel * 4));       // <= 64 KB per wave
            if (K > k_bytes) K = k_bytes;
            if (K > outer) K = outer;
            if (K < 1) K = 1;
        }
        const uint32_t groups = (outer + K - 1) / K;
        const uint32_t items = groups * C * chunks;
        hipLaunchKernelGGL(minmax_c_wave_kernel, dim3((items + kBlock / kWave - 1) / (kBlock / kWave)), dim3(kBlock), 0, s, x,
                           (uint32_t)elem_per_channel, vec_ok, C, outer, K, chunks, items, mins, maxs);
    } else {
        const int use_lds = num_channel <= 4096;
        const size_t lds = use_lds ? 2 * sizeof(float) * (size_t)num_channel : 0;
        hipLaunchKernelGGL(minmax_c_generic_kernel, dim3(stream_grid(n, kBlock * 16, num_cu() * 2)), dim3(kBlock), lds,
                           s, x, (uint32_t)n, make_fastdiv((uint32_t)elem_per_channel), nc, use_lds, mins, maxs);
    }
    return finish_launch("minmax_c");
}

int ppqhip_channel_sum(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel, double* sums,
                       void* stream) {
    if (int st = validate(n, "channel_sum")) return st;
    if (num_channel <= 0 || elem_per_channel <= 0 || n % (num_channel * elem_per_channel) != 0) {
        set_error("channel_sum: bad channel geometry"); return PPQHIP_ERR_INVALID_VALUE;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_CHANNEL_SUM, 4.0 * (double)n, s);
    const uint32_t C = (uint32_t)num_channel, epc = (uint32_t)elem_per_channel;
    const uint32_t outer = (uint32_t)(n / (num_channel * elem_per_channel));
#ifndef PPQHIP_CSUM_SINGLE
#define PPQHIP_CSUM_SINGLE 1
#endif
    if (PPQHIP_CSUM_SINGLE && epc >= 64 && epc % 4 == 0 && aligned16(x) && C >= 2u * (uint32_t)num_cu()) {
        hipLaunchKernelGGL(channel_sum_single_kernel, dim3(C), dim3(kBlock), 0, s, x, outer, C, epc, sums);
    } else if (epc >= 64) {
        uint32_t S = (4 * (uint32_t)num_cu() + C - 1) / C;            // >= 4 workgroups per CU in total
        if (S > outer) S = outer;
        if (S < 1) S = 1;
        double* partial = (double*)scratch(s, sizeof(double) * (size_t)S * C);
        if (partial == nullptr) return PPQHIP_ERR_HIP;
        const int vec_ok = (aligned16(x) && epc % 4 == 0) ? 1 : 0;
        hipLaunchKernelGGL(channel_sum_row_kernel, dim3(C, S), dim3(kBlock), 0, s, x, outer, C, epc, vec_ok, partial);
        hipLaunchKernelGGL(channel_sum_finish_kernel, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, s, partial, S,
                           C, sums);
    } else {
        hipLaunchKernelGGL(channel_sum_generic_kernel, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, s, x, outer,
                           C, epc, sums);
    }
    return finish_launch("channel_sum");
}

int ppqhip_isotone_t(const float* x, int64_t n, float* dest, void* workspace, void* stream) {
    if (int st = validate(n, "isotone_t")) return st;
    if (workspace == nullptr) { set_error("isotone_t: workspace is null"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_ISOTONE, 4.0 * (double)n, s);
    Top2* partial = (Top2*)workspace;
    const int grid = stream_grid(n, kBlock * 8, kIsotoneBlocks);
    hipLaunchKernelGGL(isotone_kernel, dim3(grid), dim3(kBlock), 0, s, x, (uint32_t)n, (const Top2*)nullptr, 0u,
                       partial, (float*)nullptr, (uint32_t)n);
    hipLaunchKernelGGL(isotone_kernel, dim3(1), dim3(kBlock), 0, s, (const float*)nullptr, 0u, (const Top2*)partial,
                       (uint32_t)grid, (Top2*)nullptr, dest, (uint32_t)n);
    return finish_launch("isotone_t");
}

int ppqhip_minmax_c_multi(const ppqhip_minmax_c_job* jobs, int num_jobs, void* stream) {
    if (num_jobs <= 0) return PPQHIP_OK;
    if (jobs == nullptr) { set_error("minmax_c_multi: jobs is null"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    double bytes = 0.0;
    for (int k = 0; k < num_jobs; k++) {
        const ppqhip_minmax_c_job& j = jobs[k];
        if (int st = validate(j.n, "minmax_c_multi")) return st;
        if (j.num_channel <= 0 || j.elem_per_channel <= 0 || j.n % (j.num_channel * j.elem_per_channel) != 0) {
            set_error("minmax_c_multi: job %d: bad channel geometry", k); return PPQHIP_ERR_INVALID_VALUE;
        }
        if (!j.x || !j.mins || !j.maxs) { set_error("minmax_c_multi: job %d has a null pointer", k); return PPQHIP_ERR_INVALID_VALUE; }
        const int64_t chunks = (j.elem_per_channel + kMMCChunk - 1) / kMMCChunk;
        if (j.fresh && (j.n != j.num_channel * j.elem_per_channel || chunks != 1)) {
            set_error("minmax_c_multi: job %d: `fresh` needs outer == 1 and elem_per_channel <= %u (one wave per channel)", k, kMMCChunk);
            return PPQHIP_ERR_INVALID_VALUE;
        }
        if ((j.n / j.elem_per_channel) * chunks > 0x7fffffffLL) { set_error("minmax_c_multi: job %d: too many rows", k); return PPQHIP_ERR_INVALID_VALUE; }
        bytes += 4.0 * (double)j.n;
    }
    LaunchScope scope(K_MINMAX_C, bytes, s);
    for (int base = 0; base < num_jobs; ) {
        MMCMultiArgs args;
        uint64_t items = 0;
        int count = 0;
        for (; count < kMMCMultiMax && base + count < num_jobs; count++) {
            const ppqhip_minmax_c_job& src = jobs[base + count];
            const uint32_t chunks = (uint32_t)((src.elem_per_channel + kMMCChunk - 1) / kMMCChunk);
            const uint64_t rows = (uint64_t)(src.n / src.elem_per_channel);
            if (items + rows * chunks > 0x7fffffffULL) {
                if (count == 0) { set_error("minmax_c_multi: too many work items in one job"); return PPQHIP_ERR_INVALID_VALUE; }
                break;                                                  // the rest goes into the next launch
            }
            args.first_item[count] = (uint32_t)items;
            items += rows * chunks;
            MMCJob& d = args.jobs[count];
            d.x = src.x; d.mins = src.mins; d.maxs = src.maxs;
            d.C = (uint32_t)src.num_channel; d.epc = (uint32_t)src.elem_per_channel;
            d.outer = (uint32_t)(rows / (uint64_t)src.num_channel); d.chunks = chunks;
            d.vec_ok = (aligned16(src.x) && src.elem_per_channel % 4 == 0) ? 1u : 0u;
            d.fresh = src.fresh ? 1u : 0u;
        }
        for (int k = count; k < kMMCMultiMax; k++) { args.first_item[k] = (uint32_t)items; args.jobs[k] = args.jobs[0]; }
        args.count = (uint32_t)count;
        args.items = (uint32_t)items;
        const uint32_t blocks = (uint32_t)((items + kBlock / kWave - 1) / (kBlock / kWave));
        hipLaunchKernelGGL(minmax_c_multi_kernel, dim3(blocks), dim3(kBlock), 0, s, args);
        base += count;
    }
    return finish_launch("minmax_c_multi");
}

}  // extern "C"
