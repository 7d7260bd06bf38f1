// reduce.hip -- range reductions and order statistics for gfx950.
//
//   minmax_t / minmax_c  one streaming pass (16-B loads), wave64 shuffle reduction, LDS across the
//                        4 waves of a workgroup, ONE pair of global atomics per workgroup on plain
//                        float storage (sign-split integer min/max).  Replaces the
//                        transpose+flatten+torch.min/max of TorchMinMaxObserver.observe
//                        (ppq/quantization/observer/range.py:86-98).
//   quantile_t           replaces Quantile_T (ppq/csrc/cuda/sort.cu:42-59): instead of a full
//                        thrust::sort of a clone it radix-SELECTS the two order statistics on the
//                        order-preserving uint32 key of the floats (12 + 12 + 8 bits, LDS histograms),
//                        no data movement; small buckets are compacted, so two streaming passes
//                        usually suffice; many tensors per launch (quantile_multi_kernel).
//   isotone_t            replaces Isotone_T (sort.cu:61-73): top-2 / bottom-2 reduction.
#include <cmath>
#include <cstdlib>

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
    uint32_t t = (uint32_t)(((uint64_t)g * jobs.total_tiles) / G);
    const uint32_t t_end = (uint32_t)(((uint64_t)(g + 1) * jobs.total_tiles) / G);
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

// rows of `epc` contiguous elements, workgroup = (row, chunk)
__global__ __launch_bounds__(kBlock) void minmax_c_row_kernel(const float* __restrict__ x, uint32_t epc, int vec_ok,
                                                              FastDiv chunks, FastDiv num_channel,
                                                              uint32_t chunk_elems, float* __restrict__ mins,
                                                              float* __restrict__ maxs) {
    __shared__ float lds[16];
    const uint32_t row = fdiv(blockIdx.x, chunks);
    const uint32_t chunk = blockIdx.x - row * chunks.d;
    const uint32_t c = row - fdiv(row, num_channel) * num_channel.d;
    const uint32_t lo = chunk * chunk_elems;
    const uint32_t hi = min(lo + chunk_elems, epc);
    const float* xr = x + (size_t)row * epc;
    float mn = INFINITY, mx = -INFINITY;
    if (vec_ok) {   // epc % 4 == 0, chunk_elems % 4 == 0, base 16-B aligned
        const float4* xv = reinterpret_cast<const float4*>(xr);
        for (uint32_t v = (lo >> 2) + threadIdx.x; v < (hi >> 2); v += kBlock) {
            const float4 a = xv[v];
            mn = fminf(fminf(mn, a.x), fminf(a.y, fminf(a.z, a.w)));
            mx = fmaxf(fmaxf(mx, a.x), fmaxf(a.y, fmaxf(a.z, a.w)));
        }
    } else {
        for (uint32_t j = lo + threadIdx.x; j < hi; j += kBlock) {
            const float a = xr[j];
            mn = fminf(mn, a); mx = fmaxf(mx, a);
        }
    }
    block_minmax_commit(mn, mx, &mins[c], &maxs[c], lds);
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
            for (uint32_t v = threadIdx.x; v < (epc >> 2); v += kBlock) {
                const float4 a = xv[v];
                acc += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
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

// ------------------------------------------------------------------------------------ quantile
// order-preserving key: ascending uint32 order == ascending float order
__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// workspace layout PER JOB (uint32 words)
constexpr int kQ1 = 4096, kQ2 = 4096, kQ3 = 256;
constexpr uint32_t kQCap = 8192;            // candidate keys kept per side when the selected bucket is small
constexpr int kOffH1 = 0;                   // hist1[4096]        : key >> 20
constexpr int kOffH2 = kOffH1 + kQ1;        // hist2[2][4096]     : (key >> 8) & 0xFFF | prefix12 match
constexpr int kOffH3 = kOffH2 + 2 * kQ2;    // hist3[2][256]      : key & 0xFF        | prefix24 match
constexpr int kOffSel = kOffH3 + 2 * kQ3;   // sel[2][8], side 0 = the q order statistic, side 1 = the (1-q) one:
enum { kSTop = 0,     // 12-bit prefix of the bucket that holds the rank
       kSRank = 1,    // rank inside that bucket
       kSMode = 2,    // kModeHist (0, after the memset) | kModeCompact | kModeDone
       kSCount = 3,   // COMPACT: candidates appended so far
       kSMin = 4,     // HIST: smallest / largest key seen in the bucket (all equal -> done after pass 2)
       kSMax = 5,
       kSP24 = 6,     // HIST, after pass 2: 24-bit prefix and the rank inside it (pass 3)
       kSR24 = 7 };
enum { kModeHist = 0, kModeCompact = 1, kModeDone = 2 };
constexpr int kOffCand = kOffSel + 16;      // cand[2][kQCap]: full keys of the bucket's elements
constexpr int kQWords = kOffCand + 2 * (int)kQCap;

// find the bin of `hist[0..nbins)` that holds rank k (0-based) and the rank inside it.
// All threads of the workgroup call this (blockDim.x == 256, nbins in {256, 4096}); thread t owns
// `per` consecutive bins, an LDS Hillis-Steele scan gives every owner its exclusive prefix and the
// one owner whose range covers k walks its (register-resident) bins.  Result: sel[0], sel[1].
__device__ void select_bin(const uint32_t* __restrict__ hist, int nbins, uint32_t k, uint32_t* scratch,
                           uint32_t* sel) {
    const int per = nbins / kBlock;   // 1 or 16
    const int t = threadIdx.x;
    uint32_t mine[16];
    uint32_t local = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        mine[j] = j < per ? hist[t * per + j] : 0u;
        local += mine[j];
    }
    __syncthreads();                  // scratch / sel may still be read from a previous call
    scratch[t] = local;
    __syncthreads();
    uint32_t incl = local;
    for (int d = 1; d < kBlock; d <<= 1) {
        const uint32_t add = t >= d ? scratch[t - d] : 0u;
        __syncthreads();
        incl += add;
        scratch[t] = incl;
        __syncthreads();
    }
    const uint32_t excl = incl - local;
    const uint32_t total = scratch[kBlock - 1];
    const uint32_t kk = k < total ? k : (total ? total - 1 : 0u);   // k < n always; guard anyway
    if (kk >= excl && kk < incl) {
        uint32_t run = excl;
        int j = 0;
#pragma unroll
        for (int jj = 0; jj < 15; jj++) {
            if (jj < per - 1 && j == jj && run + mine[jj] <= kk) { run += mine[jj]; j = jj + 1; }
        }
        sel[0] = (uint32_t)(t * per + j);
        sel[1] = kk - run;
    }
    __syncthreads();
}

constexpr int kQTrash = 64;

// Radix select of two order statistics without sorting or moving data (replaces the clone + full
// thrust::sort of Quantile_T, sort.cu:42-59), for MANY tensors per launch:
//   pass 1   (all data)   histogram of the top 12 key bits
//   select A (1 wg/job)   bucket + rank of both targets; a bucket of <= kQCap elements is COMPACTED
//   pass 2   (all data)   COMPACT: append the bucket's keys to a candidate list (a few thousand global
//                         atomics); else histogram of the middle 12 bits + min / max key of the bucket
//   select B (1 wg/job)   COMPACT: finish on the candidate list in LDS -> done.  Else: all keys equal
//                         (ReLU zeros, saturated values) -> done; otherwise 24-bit prefix for pass 3
//   pass 3   (all data)   only for sides still open (workgroups of finished jobs return at once)
//   pick     (1 wg/job)   last 8 bits
// Typical activations (0.9999 quantile: a few thousand elements in the selected bucket; the low side of
// a ReLU output: all zeros) finish after TWO passes over the data instead of three.
struct QuantileCtx {
    const float* x;
    uint32_t* ws;
    float* dest;
    uint32_t n, k_hi, k_lo;
};

__device__ __forceinline__ void quantile_pass1_body(const QuantileCtx& c, uint32_t bidx, uint32_t nblk) {
    __shared__ uint32_t h[kQ1 + kQTrash];
    for (int i = threadIdx.x; i < kQ1; i += kBlock) h[i] = 0;
    __syncthreads();
    HotCounter hc;
    hc.init(h, kQ1);
    const bool vec_ok = (reinterpret_cast<uintptr_t>(c.x) & 15u) == 0;
    stream_tiles<4>(c.x, c.n, vec_ok,
                    [&](float v, bool in) { hc.elect((int)(f2key(v) >> 20), in); },
                    [&](float v, bool in) { hc.add(in ? (int)(f2key(v) >> 20) : hc.trash()); }, bidx, nblk);
    hc.flush();
    __syncthreads();
    for (int i = threadIdx.x; i < kQ1; i += kBlock)
        if (h[i]) atomicAdd(&c.ws[kOffH1 + i], h[i]);
}

__device__ __forceinline__ void quantile_select_a_body(const QuantileCtx& c) {
    __shared__ uint32_t scratch[kBlock];
    __shared__ uint32_t sel[2];
    const uint32_t ks[2] = {c.k_hi, c.k_lo};
    for (int w = 0; w < 2; w++) {
        select_bin(c.ws + kOffH1, kQ1, ks[w], scratch, sel);
        if (threadIdx.x == 0) {
            uint32_t* S = c.ws + kOffSel + 8 * w;
            const uint32_t top = sel[0];
            S[kSTop] = top; S[kSRank] = sel[1];
            S[kSMode] = c.ws[kOffH1 + top] <= kQCap ? kModeCompact : kModeHist;
            S[kSCount] = 0; S[kSMin] = 0xFFFFFFFFu; S[kSMax] = 0u;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void quantile_pass2_body(const QuantileCtx& c, uint32_t bidx, uint32_t nblk) {
    __shared__ uint32_t h[2 * (kQ2 + kQTrash)];
    __shared__ uint32_t red[4][kBlock / kWave];
    // COMPACT sides stage their candidates in LDS and reserve their slice of the global list with ONE
    // atomic per workgroup: appending element by element would put thousands of same-address device
    // atomics (~12 ns each, serialised) on the critical path of a single large tensor
    constexpr uint32_t kLocalCap = 512;
    __shared__ uint32_t staged[2][kLocalCap];
    __shared__ uint32_t staged_n[2], staged_base[2];
    if (threadIdx.x < 2) staged_n[threadIdx.x] = 0;
    uint32_t* S_hi = c.ws + kOffSel;
    uint32_t* S_lo = c.ws + kOffSel + 8;
    const uint32_t p_hi = S_hi[kSTop], p_lo = S_lo[kSTop];
    const bool compact_hi = S_hi[kSMode] == kModeCompact, compact_lo = S_lo[kSMode] == kModeCompact;
    uint32_t* cand_hi = c.ws + kOffCand;
    uint32_t* cand_lo = c.ws + kOffCand + kQCap;
    for (int i = threadIdx.x; i < 2 * (kQ2 + kQTrash); i += kBlock) h[i] = 0;
    __syncthreads();
    HotCounter hi_c, lo_c;
    hi_c.init(h, kQ2);
    lo_c.init(h + kQ2 + kQTrash, kQ2);
    uint32_t mn_hi = 0xFFFFFFFFu, mx_hi = 0u, mn_lo = 0xFFFFFFFFu, mx_lo = 0u;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(c.x) & 15u) == 0;
    stream_tiles<4>(c.x, c.n, vec_ok,
                    [&](float v, bool in) {
                        const uint32_t key = f2key(v);
                        hi_c.elect((int)((key >> 8) & 0xFFFu), in && !compact_hi && (key >> 20) == p_hi);
                        lo_c.elect((int)((key >> 8) & 0xFFFu), in && !compact_lo && (key >> 20) == p_lo);
                    },
                    [&](float v, bool in) {
                        const uint32_t key = f2key(v);
                        const uint32_t top = key >> 20;
                        const int mid = (int)((key >> 8) & 0xFFFu);
                        if (in && top == p_hi) {
                            if (compact_hi) {
                                const uint32_t at = atomicAdd(&staged_n[0], 1u);
                                if (at < kLocalCap) staged[0][at] = key;
                                else { const uint32_t g = atomicAdd(&S_hi[kSCount], 1u); if (g < kQCap) cand_hi[g] = key; }
                            } else { hi_c.add(mid); mn_hi = min(mn_hi, key); mx_hi = max(mx_hi, key); }
                        }
                        if (in && top == p_lo) {
                            if (compact_lo) {
                                const uint32_t at = atomicAdd(&staged_n[1], 1u);
                                if (at < kLocalCap) staged[1][at] = key;
                                else { const uint32_t g = atomicAdd(&S_lo[kSCount], 1u); if (g < kQCap) cand_lo[g] = key; }
                            } else { lo_c.add(mid); mn_lo = min(mn_lo, key); mx_lo = max(mx_lo, key); }
                        }
                    }, bidx, nblk);
    hi_c.flush(); lo_c.flush();
    // workgroup min / max of the bucket keys -> one atomic pair per side
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        mn_hi = min(mn_hi, (uint32_t)__shfl_xor((int)mn_hi, m, 64)); mx_hi = max(mx_hi, (uint32_t)__shfl_xor((int)mx_hi, m, 64));
        mn_lo = min(mn_lo, (uint32_t)__shfl_xor((int)mn_lo, m, 64)); mx_lo = max(mx_lo, (uint32_t)__shfl_xor((int)mx_lo, m, 64));
    }
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wid] = mn_hi; red[1][wid] = mx_hi; red[2][wid] = mn_lo; red[3][wid] = mx_lo; }
    __syncthreads();
    if (threadIdx.x < 2) {                                         // reserve this workgroup's slice of the global lists
        const uint32_t cnt = min(staged_n[threadIdx.x], kLocalCap);
        staged_base[threadIdx.x] = cnt ? atomicAdd(&(threadIdx.x ? S_lo : S_hi)[kSCount], cnt) : 0u;
    }
    __syncthreads();
    for (int w = 0; w < 2; w++) {
        const uint32_t cnt = min(staged_n[w], kLocalCap), at = staged_base[w];
        uint32_t* cand = w ? cand_lo : cand_hi;
        for (uint32_t i = threadIdx.x; i < cnt; i += kBlock)
            if (at + i < kQCap) cand[at + i] = staged[w][i];
    }
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / kWave; w++) {
            mn_hi = min(mn_hi, red[0][w]); mx_hi = max(mx_hi, red[1][w]);
            mn_lo = min(mn_lo, red[2][w]); mx_lo = max(mx_lo, red[3][w]);
        }
        if (!compact_hi && mn_hi <= mx_hi) { atomicMin(&S_hi[kSMin], mn_hi); atomicMax(&S_hi[kSMax], mx_hi); }
        if (!compact_lo && mn_lo <= mx_lo) { atomicMin(&S_lo[kSMin], mn_lo); atomicMax(&S_lo[kSMax], mx_lo); }
    }
    for (int i = threadIdx.x; i < kQ2; i += kBlock) {
        if (!compact_hi && h[i]) atomicAdd(&c.ws[kOffH2 + i], h[i]);
        if (!compact_lo && h[kQ2 + kQTrash + i]) atomicAdd(&c.ws[kOffH2 + kQ2 + i], h[kQ2 + kQTrash + i]);
    }
}

// rank-th smallest (0-based) of cand[0..count): two LDS radix rounds over the low 20 key bits (all
// candidates share the top 12).  Every thread of the workgroup calls it; result in sel[0].
__device__ void select_in_candidates(const uint32_t* __restrict__ cand, uint32_t count, uint32_t rank, uint32_t top,
                                     uint32_t* h, uint32_t* scratch, uint32_t* sel) {
    for (int i = threadIdx.x; i < kQ2; i += kBlock) h[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < count; i += kBlock) atomicAdd(&h[(cand[i] >> 8) & 0xFFFu], 1u);
    __syncthreads();
    select_bin(h, kQ2, rank, scratch, sel);
    const uint32_t mid = sel[0], r2 = sel[1];
    __syncthreads();
    for (int i = threadIdx.x; i < kQ3; i += kBlock) h[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < count; i += kBlock)
        if (((cand[i] >> 8) & 0xFFFu) == mid) atomicAdd(&h[cand[i] & 0xFFu], 1u);
    __syncthreads();
    select_bin(h, kQ3, r2, scratch, sel);
    const uint32_t low = sel[0];
    __syncthreads();
    if (threadIdx.x == 0) sel[0] = (top << 20) | (mid << 8) | low;
    __syncthreads();
}

__device__ __forceinline__ void quantile_select_b_body(const QuantileCtx& c) {
    __shared__ uint32_t h[kQ2];
    __shared__ uint32_t scratch[kBlock];
    __shared__ uint32_t sel[2];
    for (int w = 0; w < 2; w++) {
        uint32_t* S = c.ws + kOffSel + 8 * w;
        const uint32_t mode = S[kSMode], top = S[kSTop], rank = S[kSRank];
        if (mode == kModeCompact) {
            const uint32_t count = min(S[kSCount], kQCap);
            select_in_candidates(c.ws + kOffCand + w * kQCap, count, rank, top, h, scratch, sel);
            if (threadIdx.x == 0) { c.dest[w] = key2f(sel[0]); S[kSMode] = kModeDone; }
        } else if (S[kSMin] == S[kSMax]) {                         // every element of the bucket is the same value
            if (threadIdx.x == 0) { c.dest[w] = key2f(S[kSMin]); S[kSMode] = kModeDone; }
        } else {
            select_bin(c.ws + kOffH2 + w * kQ2, kQ2, rank, scratch, sel);
            if (threadIdx.x == 0) { S[kSP24] = (top << 12) | sel[0]; S[kSR24] = sel[1]; }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void quantile_pass3_body(const QuantileCtx& c, uint32_t bidx, uint32_t nblk) {
    const uint32_t* S_hi = c.ws + kOffSel;
    const uint32_t* S_lo = c.ws + kOffSel + 8;
    const bool need_hi = S_hi[kSMode] == kModeHist, need_lo = S_lo[kSMode] == kModeHist;
    if (!need_hi && !need_lo) return;                              // the usual case: nothing left for this job
    __shared__ uint32_t h[2 * (kQ3 + kQTrash)];
    const uint32_t p_hi = S_hi[kSP24], p_lo = S_lo[kSP24];
    for (int i = threadIdx.x; i < 2 * (kQ3 + kQTrash); i += kBlock) h[i] = 0;
    __syncthreads();
    HotCounter hi_c, lo_c;
    hi_c.init(h, kQ3);
    lo_c.init(h + kQ3 + kQTrash, kQ3);
    const bool vec_ok = (reinterpret_cast<uintptr_t>(c.x) & 15u) == 0;
    stream_tiles<4>(c.x, c.n, vec_ok,
                    [&](float v, bool in) {
                        const uint32_t key = f2key(v);
                        hi_c.elect((int)(key & 0xFFu), in && need_hi && (key >> 8) == p_hi);
                        lo_c.elect((int)(key & 0xFFu), in && need_lo && (key >> 8) == p_lo);
                    },
                    [&](float v, bool in) {
                        const uint32_t key = f2key(v);
                        const int low = (int)(key & 0xFFu);
                        if (in && need_hi && (key >> 8) == p_hi) hi_c.add(low);
                        if (in && need_lo && (key >> 8) == p_lo) lo_c.add(low);
                    }, bidx, nblk);
    hi_c.flush(); lo_c.flush();
    __syncthreads();
    for (int i = threadIdx.x; i < kQ3; i += kBlock) {
        if (h[i]) atomicAdd(&c.ws[kOffH3 + i], h[i]);
        if (h[kQ3 + kQTrash + i]) atomicAdd(&c.ws[kOffH3 + kQ3 + i], h[kQ3 + kQTrash + i]);
    }
}

__device__ __forceinline__ void quantile_pick_body(const QuantileCtx& c) {
    __shared__ uint32_t scratch[kBlock];
    __shared__ uint32_t sel[2];
    for (int w = 0; w < 2; w++) {
        const uint32_t* S = c.ws + kOffSel + 8 * w;
        if (S[kSMode] == kModeHist) {
            select_bin(c.ws + kOffH3 + w * kQ3, kQ3, S[kSR24], scratch, sel);
            if (threadIdx.x == 0) c.dest[w] = key2f((S[kSP24] << 8) | sel[0]);
        }
        __syncthreads();
    }
}

// job j owns workgroups [first_block[j], first_block[j+1]) and its own kQWords-word slice of the workspace
constexpr int kQuantileMultiMax = 64;                  // jobs per launch (2.6 KB of kernel arguments)
constexpr uint32_t kQuantileMultiChunk = 32u << 10;    // elements per workgroup (128 KB)
constexpr uint32_t kQuantileMultiCap = 1024;           // workgroups per job at most
struct QuantileJob {
    const float* x;
    uint32_t* ws;
    float* dest;
    uint32_t n, k_hi, k_lo, first_block;
};
struct QuantileJobs {
    QuantileJob job[kQuantileMultiMax];
    uint32_t count;
};
enum { kQPass1 = 1, kQSelectA, kQPass2, kQSelectB, kQPass3, kQPick };

template <int STEP>
__global__ __launch_bounds__(kBlock) void quantile_multi_kernel(const QuantileJobs jobs) {
    constexpr bool per_job = STEP == kQSelectA || STEP == kQSelectB || STEP == kQPick;     // one workgroup per job
    uint32_t lo = per_job ? blockIdx.x : 0, hi = jobs.count;
    while (!per_job && hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs.job[mid].first_block <= blockIdx.x) lo = mid; else hi = mid;
    }
    const QuantileJob& j = jobs.job[lo];
    QuantileCtx c;
    c.x = j.x; c.ws = j.ws; c.dest = j.dest; c.n = j.n; c.k_hi = j.k_hi; c.k_lo = j.k_lo;
    const uint32_t end = lo + 1 < jobs.count ? jobs.job[lo + 1].first_block : gridDim.x;
    const uint32_t bidx = blockIdx.x - j.first_block, nblk = end - j.first_block;
    if (STEP == kQPass1) quantile_pass1_body(c, bidx, nblk);
    if (STEP == kQSelectA) quantile_select_a_body(c);
    if (STEP == kQPass2) quantile_pass2_body(c, bidx, nblk);
    if (STEP == kQSelectB) quantile_select_b_body(c);
    if (STEP == kQPass3) quantile_pass3_body(c, bidx, nblk);
    if (STEP == kQPick) quantile_pick_body(c);
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

static int validate(int64_t n, const char* what) {
    if (n <= 0) { set_error("%s: tensor is empty", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (n > 0x7fffffffLL) { set_error("%s: too many elements", what); return PPQHIP_ERR_INVALID_VALUE; }
    return PPQHIP_OK;
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
    const int grid = stream_grid(n, kBlock * 4 * 4, kNumCU * 8);
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
    if (grid > (uint32_t)kMMGrid) grid = kMMGrid;
    // streaming (nontemporal) loads once the data cannot be cache resident
    if (elems >= (48ll << 20)) hipLaunchKernelGGL((minmax_persistent_kernel<true>), dim3(grid), dim3(kMMBlock), 0, s, args);
    else hipLaunchKernelGGL((minmax_persistent_kernel<false>), dim3(grid), dim3(kMMBlock), 0, s, args);
}

int ppqhip_minmax_t_slots(const float* x, int64_t n, float* slots, void* stream) {
    if (slots == nullptr) { set_error("minmax_t_slots: slots is null"); return PPQHIP_ERR_INVALID_VALUE; }
    if (int st = validate(n, "minmax_t_slots")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_MINMAX_T, 4.0 * (double)n, s);
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
        const uint32_t chunk_elems = 8192;
        const uint32_t chunks = (uint32_t)((elem_per_channel + chunk_elems - 1) / chunk_elems);
        const int64_t rows = n / elem_per_channel;
        const int vec_ok = (aligned16(x) && elem_per_channel % 4 == 0) ? 1 : 0;
        hipLaunchKernelGGL(minmax_c_row_kernel, dim3((uint32_t)(rows * chunks)), dim3(kBlock), 0, s, x,
                           (uint32_t)elem_per_channel, vec_ok, make_fastdiv(chunks), nc, chunk_elems, mins, maxs);
    } else {
        const int use_lds = num_channel <= 4096;
        const size_t lds = use_lds ? 2 * sizeof(float) * (size_t)num_channel : 0;
        hipLaunchKernelGGL(minmax_c_generic_kernel, dim3(stream_grid(n, kBlock * 16, kNumCU * 2)), dim3(kBlock), lds,
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
    if (epc >= 64) {
        uint32_t S = (4 * kNumCU + C - 1) / C;            // >= 4 workgroups per CU in total
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

int64_t ppqhip_quantile_workspace_bytes(int64_t n) {
    (void)n;
    const int64_t q = (int64_t)kQWords * 4;
    const int64_t iso = (int64_t)kIsotoneBlocks * (int64_t)sizeof(Top2);
    return q > iso ? q : iso;
}

static int quantile_multi_impl(const ppqhip_quantile_job* jobs, int num_jobs, float q, void* workspace, hipStream_t s,
                               const char* what) {
    uint32_t* ws = (uint32_t*)workspace;
    if (int st = check_hip(hipMemsetAsync(ws, 0, (size_t)num_jobs * kQWords * 4, s), "memset quantile workspace"))
        return st;
    for (int base = 0; base < num_jobs; base += kQuantileMultiMax) {
        QuantileJobs args;
        args.count = (uint32_t)((num_jobs - base) < kQuantileMultiMax ? (num_jobs - base) : kQuantileMultiMax);
        uint32_t blocks = 0;
        for (uint32_t k = 0; k < args.count; k++) {
            const ppqhip_quantile_job& src = jobs[base + k];
            const int64_t n = src.n;
            // index rule of _Quantile_T, sort.cu:13-19: __float2int_rn(num_of_elements * q), clipped to [0, n-1]
            auto pos = [n](float f) -> uint32_t {
                float p = nearbyintf((float)n * f);
                if (!(p > 0.f)) return 0u;                      // also NaN
                if (p >= (float)(n - 1)) return (uint32_t)(n - 1);
                return (uint32_t)p;
            };
            QuantileJob& d = args.job[k];
            d.x = src.x; d.dest = src.dest; d.n = (uint32_t)n; d.ws = ws + (size_t)(base + k) * kQWords;
            d.k_hi = pos(q); d.k_lo = pos(1 - q); d.first_block = blocks;
            uint32_t nb = (uint32_t)((n + kQuantileMultiChunk - 1) / kQuantileMultiChunk);
            if (nb > kQuantileMultiCap) nb = kQuantileMultiCap;
            if (nb < 1) nb = 1;
            blocks += nb;
        }
        const dim3 all(blocks), one(args.count), wg(kBlock);
        hipLaunchKernelGGL(quantile_multi_kernel<kQPass1>, all, wg, 0, s, args);
        hipLaunchKernelGGL(quantile_multi_kernel<kQSelectA>, one, wg, 0, s, args);
        hipLaunchKernelGGL(quantile_multi_kernel<kQPass2>, all, wg, 0, s, args);
        hipLaunchKernelGGL(quantile_multi_kernel<kQSelectB>, one, wg, 0, s, args);
        hipLaunchKernelGGL(quantile_multi_kernel<kQPass3>, all, wg, 0, s, args);
        hipLaunchKernelGGL(quantile_multi_kernel<kQPick>, one, wg, 0, s, args);
    }
    return finish_launch(what);
}

int ppqhip_quantile_t(const float* x, int64_t n, float q, float* dest, void* workspace, void* stream) {
    if (int st = validate(n, "quantile_t")) return st;
    if (workspace == nullptr) { set_error("quantile_t: workspace is null"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_QUANTILE, 4.0 * (double)n, s);
    ppqhip_quantile_job job;
    job.x = x; job.dest = dest; job.n = n;
    return quantile_multi_impl(&job, 1, q, workspace, s, "quantile_t");
}

int64_t ppqhip_quantile_multi_workspace_bytes(int num_jobs) {
    return num_jobs > 0 ? (int64_t)num_jobs * kQWords * 4 : 0;
}

int ppqhip_quantile_t_multi(const ppqhip_quantile_job* jobs, int num_jobs, float q, void* workspace, void* stream) {
    if (num_jobs <= 0) return PPQHIP_OK;
    if (jobs == nullptr || workspace == nullptr) {
        set_error("quantile_t_multi: jobs / workspace is null"); return PPQHIP_ERR_INVALID_VALUE;
    }
    double bytes = 0.0;
    for (int k = 0; k < num_jobs; k++) {
        if (int st = validate(jobs[k].n, "quantile_t_multi")) return st;
        if (jobs[k].x == nullptr || jobs[k].dest == nullptr) {
            set_error("quantile_t_multi: job %d has a null pointer", k); return PPQHIP_ERR_INVALID_VALUE;
        }
        bytes += 4.0 * (double)jobs[k].n;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_QUANTILE, bytes, s);
    return quantile_multi_impl(jobs, num_jobs, q, workspace, s, "quantile_t_multi");
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

}  // extern "C"
