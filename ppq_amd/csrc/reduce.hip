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
// (an explicit unsigned max: `max` resolves to the int overload in the host pass of this translation unit)
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
    return v;
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
constexpr int kOffH0 = kOffCand + 2 * (int)kQCap;   // hist0[4096]: key >> 20 of the SAMPLE (speculation, see below)
constexpr int kOffR0 = kOffH0 + kQ1;        // round0[4096]: those of them that ARE their bucket's round key (0, 6.0, -1.0 ..)
constexpr int kOffSpec = kOffR0 + kQ1;      // spec[16]: thresholds + counters of the speculative lists
enum { kPEnabled = 0,  // 1 once select0 has chosen thresholds
       kPTHi = 1,      // keys > T_hi are appended to the hi list (0xFFFFFFFF: none)
       kPTLo = 2,      // keys < T_lo are appended to the lo list (0: none);  T_lo <= T_hi
       kPCntHi = 3, kPCntLo = 4,          // keys appended (may exceed the capacity: then the list is unusable)
       kPTieHi = 5, kPTieLo = 6,          // LOWER BOUNDS of the number of keys == T_hi / == T_lo (one element in eight is looked at)
       kPOvfHi = 9, kPOvfLo = 10 };       // some workgroup met more matching keys than it can stage: list incomplete
constexpr int kQWords = kOffSpec + 16;
// the key of the smallest-magnitude value of bucket b (3 mantissa bits): the values activations TIE on -- 0 after a ReLU, 6.0
// after a ReLU6 / clip, +-1 after a saturating function -- are of this form
__host__ __device__ inline uint32_t round_key_of_bucket(uint32_t b) { return b >= 0x800u ? (b << 20) : ((b << 20) | 0xFFFFFu); }
// capacity (keys per side) of a job's speculative lists; they live behind the fixed parts of all jobs
__host__ __device__ inline uint32_t quantile_spec_cap(uint64_t n) {
    uint64_t c = n / 128;
    if (c < 4096) c = 4096;
    if (c > (1u << 20)) c = 1u << 20;
    return (uint32_t)((c + 3) & ~3ull);             // lists stay 16-B aligned
}
constexpr uint32_t kQSampleVec = 256;       // float4 sampled at the head of every workgroup's chunk (one per lane)
#ifndef PPQHIP_Q_SPEC_MIN_ELEMS
#define PPQHIP_Q_SPEC_MIN_ELEMS (1ll << 18)
#endif
constexpr int64_t kQSpeculateMinElems = PPQHIP_Q_SPEC_MIN_ELEMS;   // smaller launches skip the speculation (two launches saved)

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
//   pass 1   (all data)      histogram of the top 12 key bits (persistent kernel, below)
//   select A (1 wg/job/side) bucket + rank of the target; a bucket of <= kQCap elements is COMPACTED
//   pass 2   (all data)      COMPACT: append the bucket's keys to a candidate list (a few thousand global
//                            atomics); else histogram of the middle 12 bits + min / max key of the bucket
//   select B (1 wg/job)      COMPACT: finish on the candidate list in LDS -> done.  Else: all keys equal
//                            (saturated values) -> done; otherwise 24-bit prefix for pass 3
//   pass 3   (all data)      only for sides still open (workgroups of finished jobs return at once)
//   pick     (1 wg/job)      last 8 bits
// (Folding the single-workgroup steps into the tail of the pass before them -- "last block done" -- was tried: the
// agent-scope release / acquire it needs is an L2 write-back + invalidate per workgroup on this 8-XCD part, and the
// sample launch went from 10 to 77 us.  Launch boundaries are the cheaper fence.)
//
// Speculation (round 2): for the extreme order statistics calibration asks for (q = 0.9999) the answer lies among a
// few thousand elements, and reading the whole tensor again just to look at them is what kept this kernel at
// ~0.5 passes^-1 of the roofline.  A SAMPLE launch (64 jittered 64-B granules per 128 KB chunk, ~1-3 % of the data)
// histograms the top 12 key bits, and how many of each bucket's keys are the bucket's round key; select0
// picks KEY thresholds T_hi / T_lo such that ~1.3-2x the wanted number of elements lies beyond them; pass 1
// then, while building the exact histogram, STAGES every key beyond a threshold for a list (one compare per float4 in
// the streaming loop, ~0.03 % of the elements take the branch) and counts a lower bound of the keys EQUAL to a
// threshold (ties: ReLU zeros, ReLU6 sixes).  Select A finds the wanted rank in the list or proves it is the tied
// threshold value; only if neither holds (unlucky sample, overflow, ties on an odd value) passes 2 / 3 run as before.
// The result is exact either way; in the usual case the tensor is read ONCE (+1 % for the sample).
struct QuantileCtx {
    const float* x;
    uint32_t* ws;
    uint32_t* spec;       // [2][cap] speculative key lists (hi, lo)
    float* dest;
    uint32_t n, k_hi, k_lo, cap;
};

// thresholds from the sample histogram: one workgroup per job.
// hi side: bucket b = the LARGEST with (sample count of top >= b) >= need; the threshold lies INSIDE it -- at its round key
// when at least half of the bucket's sample is that one value (ties), else interpolated so that about 1.6x the still missing
// count lies above it (the density falls towards the extreme, a linear share would come up short).  lo side mirrored.
__device__ __forceinline__ void quantile_select0_body(const QuantileCtx& c) {
    __shared__ uint32_t scratch[kBlock];
    __shared__ uint32_t stop[2], thr[2];
    const int t = threadIdx.x;
    constexpr int per = kQ1 / kBlock;                  // 16 bins per thread
    uint32_t mine[per];
    uint32_t local = 0;
#pragma unroll
    for (int j = 0; j < per; j++) { mine[j] = c.ws[kOffH0 + t * per + j]; local += mine[j]; }
    if (t == 0) { stop[0] = 0u; stop[1] = kQ1; thr[0] = 0xFFFFFFFFu; thr[1] = 0u; }
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
    const uint32_t m = scratch[kBlock - 1];            // sample size
    if (m == 0) return;                                // spec[kPEnabled] stays 0
    const double frac = (double)m / (double)c.n;
    // 1.3x the expected sample count + 16: a too small list is caught by the exact histogram (fallback), never wrong
    const double need_hi = 1.3 * frac * (double)(c.n - 1 - c.k_hi) + 16.0;
    const double need_lo = 1.3 * frac * (double)c.k_lo + 16.0;
    const double budget = frac * (double)(c.cap / 2);
    uint32_t F = incl - local;                         // F(b) = sample count with top < b, here b = t * per
    uint32_t best_hi = 0, best_lo = kQ1;
    bool any_hi = false;
#pragma unroll
    for (int j = 0; j < per; j++) {
        const uint32_t b = (uint32_t)(t * per + j);
        const uint32_t Fb = F, Fb1 = F + mine[j];
        if ((double)Fb1 >= need_lo && b < best_lo) best_lo = b;
        if ((double)(m - Fb) >= need_hi) { best_hi = b; any_hi = true; }
        F = Fb1;
    }
    if (best_lo < kQ1) atomicMin(&stop[1], best_lo);
    if (any_hi) atomicMax(&stop[0], best_hi);
    __syncthreads();
    const uint32_t bh = stop[0], bl = stop[1];
    const bool have_hi = (double)m >= need_hi, have_lo = bl < kQ1;
    F = incl - local;
#pragma unroll
    for (int j = 0; j < per; j++) {
        const uint32_t b = (uint32_t)(t * per + j);
        const uint32_t Fb = F, Fb1 = F + mine[j];
        const uint32_t L = b << 20, H = L | 0xFFFFFu, R = round_key_of_bucket(b);
        const double cnt = (double)mine[j];
        if (have_hi && b == bh) {
            const double above = (double)(m - Fb1), missing = need_hi - above;          // missing in (0, cnt]
            const double round = (double)c.ws[kOffR0 + b];
            uint32_t T = 0xFFFFFFFFu;
            if (2.0 * round >= cnt) {
                if (above + (R == L ? cnt - round : 0.0) <= budget) T = R;
            } else {
                const double take = fmin(cnt, 1.6 * missing);
                if (above + take <= budget) {
                    const uint32_t w = (uint32_t)(take / cnt * 1048576.0);
                    T = w >= 0x100000u ? (L ? L - 1u : 0u) : H - w;
                }
            }
            thr[0] = T;
        }
        if (have_lo && b == bl) {
            const double below = (double)Fb, missing = need_lo - below;
            const double round = (double)c.ws[kOffR0 + b];
            uint32_t T = 0u;
            if (2.0 * round >= cnt) {
                if (below + (R == H ? cnt - round : 0.0) <= budget) T = R;
            } else {
                const double take = fmin(cnt, 1.6 * missing);
                if (below + take <= budget) {
                    const uint32_t w = (uint32_t)(take / cnt * 1048576.0);
                    T = w >= 0x100000u ? (H == 0xFFFFFFFFu ? H : H + 1u) : L + w;
                }
            }
            thr[1] = T;
        }
        F = Fb1;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t* P = c.ws + kOffSpec;
        if (thr[1] > thr[0]) return;                   // thresholds cross (tiny / degenerate sample): no speculation
        P[kPTHi] = thr[0]; P[kPTLo] = thr[1];
        P[kPCntHi] = 0; P[kPCntLo] = 0; P[kPTieHi] = 0; P[kPTieLo] = 0; P[kPOvfHi] = 0; P[kPOvfLo] = 0;
        P[kPEnabled] = 1u;
    }
}

#ifndef PPQHIP_QS_STRIDE
#define PPQHIP_QS_STRIDE 4
#endif
// every 4th workgroup samples the heads of 4 chunks: the launch is bound by the LDS atomics of ONE workgroup
// (measured on [32,512,56,56], randn / relu: stride 1: 24 / 22 us, 2: 14 / 15, 4: 11 / 14, 8: 12 / 22, 16: 19 / 38)
constexpr uint32_t kQSampleStride = PPQHIP_QS_STRIDE;
__device__ __forceinline__ void quantile_sample_body(const QuantileCtx& c, uint32_t bidx, uint32_t nblk) {
    __shared__ uint32_t h[kQ1], hr[kQ1];
    const bool vec_ok = (reinterpret_cast<uintptr_t>(c.x) & 15u) == 0;
    if (!vec_ok || bidx % kQSampleStride != 0) return;     // unaligned tensors: no sample -> no speculation
    for (int i = threadIdx.x; i < kQ1; i += kBlock) { h[i] = 0; hr[i] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // every lane of the wave calls this; when >= 16 lanes share the first lane's bucket they count with ONE ds_add (after a
    // ReLU half of the sample is the same key, and 256 same-address LDS atomics per value made this launch 3x longer)
    auto count = [&](float f, bool valid) {
        const uint32_t key = f2key(f), top = key >> 20;
        const bool round = key == round_key_of_bucket(top);
        const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)top);
        const bool same = valid && top == lead;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(same);
        if (__builtin_popcountll(m) < 16) {            // wave-uniform: no tie worth aggregating
            if (valid) { atomicAdd(&h[top], 1u); if (round) atomicAdd(&hr[top], 1u); }
            return;
        }
        const unsigned long long mr = __builtin_amdgcn_ballot_w64(same && round);
        if (same) {
            if (lane == __builtin_ctzll(m)) {
                atomicAdd(&h[top], (uint32_t)__builtin_popcountll(m));
                if (mr) atomicAdd(&hr[top], (uint32_t)__builtin_popcountll(mr));
            }
        } else if (valid) {
            atomicAdd(&h[top], 1u);
            if (round) atomicAdd(&hr[top], 1u);
        }
    };
    // the same chunking as stream_tiles<4>: workgroup b owns float4 [b * per * tile, (b + 1) * per * tile)
    const uint32_t nvec = c.n >> 2, tile = kBlock * 4;
    const uint32_t tiles = (nvec + tile - 1) / tile;
    const uint32_t per = (tiles + nblk - 1) / nblk;
    float4 a[kQSampleStride];
    bool ok[kQSampleStride];
    // 64 granules of 64 B (4 float4: one memory sector each) per chunk, one every chunk / 64 with a hashed offset inside
    // its window -- NOT the contiguous head of the chunk: activations are channel-structured ([N, C, H, W]; the extreme
    // quantile lives in a few channels), a contiguous 4 KB run sees one channel's rows and on real networks the
    // thresholds came out wrong often enough to send half of the data through the fall-back passes (ResNet-50, 72
    // tensors: 978 us per forward); the jitter breaks any period the channel stride shares with the window.
    const uint32_t chunk_vec = per * tile, window = chunk_vec / 64u, granule = threadIdx.x >> 2, sub = threadIdx.x & 3u;
#pragma unroll
    for (uint32_t j = 0; j < kQSampleStride; j++) {
        const uint32_t lo = (bidx + j) * chunk_vec;
        uint32_t v = lo + threadIdx.x;                                  // tiny chunks: the contiguous head
        if (window >= 8u) {
            const uint32_t slots = window / 4u;                         // 64-B aligned positions inside the window
            const uint32_t h = ((granule * 2654435761u) ^ ((bidx + j) * 40503u + 0x9E3779B9u)) >> 9;
            v = lo + granule * window + (h % slots) * 4u + sub;
        }
        ok[j] = bidx + j < nblk && threadIdx.x < kQSampleVec && v < nvec && v < lo + chunk_vec;
        a[j] = reinterpret_cast<const float4*>(c.x)[ok[j] ? v : 0u];
    }
#pragma unroll
    for (uint32_t j = 0; j < kQSampleStride; j++) {
        count(a[j].x, ok[j]); count(a[j].y, ok[j]); count(a[j].z, ok[j]); count(a[j].w, ok[j]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kQ1; i += kBlock) {
        if (h[i]) atomicAdd(&c.ws[kOffH0 + i], h[i]);
        if (hr[i]) atomicAdd(&c.ws[kOffR0 + i], hr[i]);
    }
}

// Selection inside a key list (16-B aligned), restricted to the keys whose top 12 bits equal `top`: two radix rounds
// (12 + 8 bits) with the histogram in LDS, 16-B loads, 8 of them in flight per lane.  Every thread of the workgroup
// calls these.  list_round1 builds the first histogram in h[0..kQ2) and returns how many keys matched;
// list_finish returns (in sel[0]) the rank-th smallest of them (0-based, rank < matched).
template <typename F>
__device__ __forceinline__ void list_sweep(const uint32_t* __restrict__ list, uint32_t count, F&& f) {
    const uint4* lv = reinterpret_cast<const uint4*>(list);
    const uint32_t nv = (count + 3) >> 2;
    for (uint32_t i = threadIdx.x; i < nv; i += 8 * kBlock) {
        uint4 k[8];
#pragma unroll
        for (int u = 0; u < 8; u++) k[u] = lv[min(i + u * kBlock, nv - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t at = (i + u * kBlock) << 2;
            if (i + u * kBlock < nv) {
                if (at + 0 < count) f(k[u].x);
                if (at + 1 < count) f(k[u].y);
                if (at + 2 < count) f(k[u].z);
                if (at + 3 < count) f(k[u].w);
            }
        }
    }
}
__device__ uint32_t list_round1(const uint32_t* __restrict__ list, uint32_t count, uint32_t top, uint32_t* h, uint32_t* scratch) {
    for (int i = threadIdx.x; i < kQ2; i += kBlock) h[i] = 0;
    if (threadIdx.x == 0) scratch[0] = 0;
    __syncthreads();
    uint32_t mine = 0;
    list_sweep(list, count, [&](uint32_t key) { if ((key >> 20) == top) { atomicAdd(&h[(key >> 8) & 0xFFFu], 1u); mine++; } });
    mine = wave_sum_u32(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&scratch[0], mine);
    __syncthreads();
    const uint32_t matched = scratch[0];
    __syncthreads();
    return matched;
}
__device__ void list_finish(const uint32_t* __restrict__ list, uint32_t count, uint32_t top, uint32_t rank, uint32_t* h,
                            uint32_t* scratch, uint32_t* sel) {
    select_bin(h, kQ2, rank, scratch, sel);
    const uint32_t mid = sel[0], r2 = sel[1];
    __syncthreads();
    for (int i = threadIdx.x; i < kQ3; i += kBlock) h[i] = 0;
    __syncthreads();
    const uint32_t p24 = (top << 12) | mid;
    list_sweep(list, count, [&](uint32_t key) { if ((key >> 8) == p24) atomicAdd(&h[key & 0xFFu], 1u); });
    __syncthreads();
    select_bin(h, kQ3, r2, scratch, sel);
    const uint32_t low = sel[0];
    __syncthreads();
    if (threadIdx.x == 0) sel[0] = (p24 << 8) | low;
    __syncthreads();
}
__device__ void select_in_list(const uint32_t* __restrict__ list, uint32_t count, uint32_t top, uint32_t rank, uint32_t* h,
                               uint32_t* scratch, uint32_t* sel) {
    list_round1(list, count, top, h, scratch);
    list_finish(list, count, top, rank, h, scratch, sel);
}

// one workgroup per (job, side).  With the exact histogram: bucket `top` of the wanted rank and the rank r inside it.
// The speculative list of the side holds EVERY key beyond the threshold T (unless it overflowed), i.e. the `listed`
// most extreme keys of bucket `top`; directly inside of them lie the >= tie keys equal to T (T's own bucket only).
//   hi:  r >= in_bucket - listed -> the (r - (in_bucket - listed))-th smallest listed key;  else within `tie` of it -> T
//   lo:  r < listed -> the r-th smallest listed key;  else r - listed < tie -> T
// Anything else (unlucky sample, overflow, a tie on a value that is not a round key) falls back to passes 2 / 3.
__device__ __forceinline__ void quantile_select_a_body(const QuantileCtx& c, int w) {
    __shared__ uint32_t scratch[kBlock];
    __shared__ uint32_t sel[2];
    __shared__ uint32_t h[kQ1];
    const uint32_t* P = c.ws + kOffSpec;
    select_bin(c.ws + kOffH1, kQ1, w ? c.k_lo : c.k_hi, scratch, sel);
    const uint32_t top = sel[0], rank = sel[1];
    __syncthreads();
    bool done = false;
    const uint32_t count = P[w ? kPCntLo : kPCntHi];
    if (P[kPEnabled] != 0u && count <= c.cap && P[w ? kPOvfLo : kPOvfHi] == 0u) {
        const uint32_t T = P[w ? kPTLo : kPTHi], tie = P[w ? kPTieLo : kPTieHi];
        const uint32_t* list = c.spec + (w ? c.cap : 0u);
        const uint32_t in_bucket = c.ws[kOffH1 + top];
        const uint32_t listed = list_round1(list, count, top, h, scratch);
        const uint32_t first_listed = w ? 0u : in_bucket - listed;       // ranks [first_listed, first_listed + listed) are listed
        if (rank >= first_listed && rank - first_listed < listed) {
            list_finish(list, count, top, rank - first_listed, h, scratch, sel);
            if (threadIdx.x == 0) c.dest[w] = key2f(sel[0]);
            done = true;
        } else if ((T >> 20) == top) {
            const uint32_t away = w ? rank - listed + 1u : first_listed - rank;      // 1 = the key next to the listed ones
            if (away >= 1u && away <= tie) {
                if (threadIdx.x == 0) c.dest[w] = key2f(T);
                done = true;
            }
        }
    }
    if (threadIdx.x == 0) {
        uint32_t* S = c.ws + kOffSel + 8 * w;
        S[kSTop] = top; S[kSRank] = rank;
        S[kSMode] = done ? kModeDone : (c.ws[kOffH1 + top] <= kQCap ? kModeCompact : kModeHist);
        S[kSCount] = 0; S[kSMin] = 0xFFFFFFFFu; S[kSMax] = 0u;
    }
}

__device__ __forceinline__ void quantile_select_b_body(const QuantileCtx& c) {
    __shared__ uint32_t h[kQ2];
    __shared__ uint32_t scratch[kBlock];
    __shared__ uint32_t sel[2];
    for (int w = 0; w < 2; w++) {
        uint32_t* S = c.ws + kOffSel + 8 * w;
        const uint32_t mode = S[kSMode], top = S[kSTop], rank = S[kSRank];
        if (mode == kModeDone) {
        } else if (mode == kModeCompact) {
            const uint32_t count = min(S[kSCount], kQCap);
            select_in_list(c.ws + kOffCand + w * kQCap, count, top, rank, h, scratch, sel);
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
    if (S_hi[kSMode] == kModeDone && S_lo[kSMode] == kModeDone) return;       // the speculation settled both sides
    // a finished side must match nothing: 0xFFFFFFFF is no 12-bit prefix
    const uint32_t p_hi = S_hi[kSMode] == kModeDone ? 0xFFFFFFFFu : S_hi[kSTop];
    const uint32_t p_lo = S_lo[kSMode] == kModeDone ? 0xFFFFFFFFu : S_lo[kSTop];
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

// job j owns workgroups [first_block[j], first_block[j+1]) and its own kQWords-word slice of the workspace
constexpr int kQuantileMultiMax = 64;                  // jobs per launch (2.6 KB of kernel arguments)
constexpr uint32_t kQuantileMultiChunk = 32u << 10;    // elements per workgroup (128 KB)
constexpr uint32_t kQuantileMultiCap = 1024;           // workgroups per job at most
struct QuantileJob {
    const float* x;
    uint32_t* ws;
    uint32_t* spec;
    float* dest;
    uint32_t n, k_hi, k_lo, first_block, cap, first_tile;
};
struct QuantileJobs {
    QuantileJob job[kQuantileMultiMax];
    uint32_t count, total_tiles;
};
enum { kQSelectA = 2, kQPass2, kQSelectB, kQPass3, kQPick, kQSample, kQSelect0 };

// ---- pass 1 as a persistent kernel (the design of hist_persistent_kernel, hist.hip) ----------------------
// The work is the concatenated list of tiles (kQ1Block * kQ1U float4) of all jobs, split evenly over a chip-sized
// grid; a workgroup walks its contiguous range with two ping-pong register tiles, counts key >> 20 in ONE LDS
// histogram (EXEC-mask commits + hot bin: WaveBinCounter) and, per job it touches, adds the non-zero bins to that
// job's hist1 with device atomics.  SPEC: keys outside (B_lo, B_hi) are staged for the speculative lists; the test
// is one subtract + max per element and ONE wave-level branch per float4 (the element-wise branches of the first
// version cost as much as a second pass: 13 SALU / element).
constexpr int kQ1Block = 512, kQ1U = 2, kQ1WgPerCu = 2;
constexpr uint32_t kQ1TileVec = kQ1Block * kQ1U, kQ1TileElems = kQ1TileVec * 4;
constexpr uint32_t kQ1LocalCap = 1024;            // keys a workgroup can stage per side and job
#ifndef PPQHIP_Q1_COPIES
#define PPQHIP_Q1_COPIES 4
#endif
constexpr int kQ1Copies = PPQHIP_Q1_COPIES;
__host__ __device__ inline uint32_t q1_job_tiles(uint32_t n, bool vec_ok) {
    if (!vec_ok) return (n + kQ1TileElems - 1) / kQ1TileElems;
    const uint32_t full = (n >> 2) / kQ1TileVec;
    return full + (n > full * kQ1TileElems ? 1u : 0u);
}

// A key outside [T_lo, T_hi]: stage it for the speculative list of its side.  Deliberately NOT inlined (eight inlined
// copies tripled the streaming loop's code size for a branch ~5 % of the wave iterations take).
__device__ __noinline__ void q1_rare_key(uint32_t key, uint32_t t_hi, uint32_t* staged_hi, uint32_t* staged_lo,
                                         uint32_t* staged_n) {
    const int w = key > t_hi ? 0 : 1;
    const uint32_t at = atomicAdd(&staged_n[w], 1u);
    if (at < kQ1LocalCap) (w ? staged_lo : staged_hi)[at] = key;
}

template <bool SPEC>
__global__ __launch_bounds__(kQ1Block, (kQ1Block * kQ1WgPerCu + 255) / 256)
void quantile_pass1_persistent_kernel(const QuantileJobs jobs) {
    // kQ1Copies histogram copies, chosen by lane: activations put most keys into a few dozen exponent buckets, and a
    // k-way same-address ds_add costs ~k cycles -- spreading the lanes of a wave over 4 copies cuts the conflicts 4x
    __shared__ int h[kQ1Copies * kQ1];
    __shared__ uint32_t staged[2][SPEC ? kQ1LocalCap : 1];
    __shared__ uint32_t staged_n[2], staged_base[2];
    __shared__ uint32_t ties[2];                  // keys seen == T_hi / == T_lo (this workgroup, this job)
    const uint32_t G = gridDim.x, g = blockIdx.x;
    uint32_t t = (uint32_t)(((uint64_t)g * jobs.total_tiles) / G);
    const uint32_t t_end = (uint32_t)(((uint64_t)(g + 1) * jobs.total_tiles) / G);
    if (t >= t_end) return;
    for (int i = threadIdx.x; i < kQ1Copies * kQ1; i += kQ1Block) h[i] = 0;
    if (threadIdx.x < 2) { staged_n[threadIdx.x] = 0; ties[threadIdx.x] = 0; }
    __syncthreads();
    uint32_t lo = 0, hi = jobs.count;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs.job[mid].first_tile <= t) lo = mid; else hi = mid;
    }
    WaveBinCounter<false, true, true> acc;
    acc.init(h + (threadIdx.x % kQ1Copies) * kQ1, kQ1);
    for (uint32_t j = lo; t < t_end; j++) {
        const QuantileJob& job = jobs.job[j];
        const uint32_t j_end = (j + 1 < jobs.count) ? jobs.job[j + 1].first_tile : jobs.total_tiles;
        uint32_t k = t - job.first_tile;
        const uint32_t k1 = min(t_end, j_end) - job.first_tile;
        t = min(t_end, j_end);
        const float* __restrict__ x = job.x;
        const uint32_t n = job.n;
        uint32_t* P = job.ws + kOffSpec;
        const bool spec = SPEC && P[kPEnabled] != 0u;
        // not enabled: [0, 0xFFFFFFFF] has no outside
        const uint32_t t_hi = spec ? P[kPTHi] : 0xFFFFFFFFu, t_lo = spec ? P[kPTLo] : 0u;
        const uint32_t span = t_hi - t_lo;                             // t_lo <= t_hi always (select0); key - t_lo > span <=> outside
        int tie_hi = 0, tie_lo = 0;                                    // wave-uniform
        auto rare = [&](uint32_t key) { q1_rare_key(key, t_hi, staged[0], staged[1], staged_n); };
        const bool vec_ok = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
        const uint32_t full = vec_ok ? (n >> 2) / kQ1TileVec : 0u;
        const uint32_t kf = min(k1, full);
        if (k < kf) {
            const float4* xv = reinterpret_cast<const float4*>(x) + threadIdx.x;
            float4 bufa[kQ1U], bufb[kQ1U];
            auto fetch = [&](float4 (&buf)[kQ1U], uint32_t tile) {
                const float4* p = xv + (size_t)tile * kQ1TileVec;
#pragma unroll
                for (int u = 0; u < kQ1U; u++) buf[u] = load4<true>(p + u * kQ1Block);
            };
            auto consume = [&](const float4 (&buf)[kQ1U]) {
#pragma unroll
                for (int u = 0; u < kQ1U; u++) {
                    const uint32_t k0 = f2key(buf[u].x), k1_ = f2key(buf[u].y), k2 = f2key(buf[u].z), k3 = f2key(buf[u].w);
                    int b[4] = {(int)(k0 >> 20), (int)(k1_ >> 20), (int)(k2 >> 20), (int)(k3 >> 20)};
                    if (u == 0) acc.elect(b[0], true);
                    acc.commit4_exec(b);
                    if (SPEC) {
                        const uint32_t d0 = k0 - t_lo, d1 = k1_ - t_lo, d2 = k2 - t_lo, d3 = k3 - t_lo;
                        if (u == 0) {      // ties on the thresholds: a lower bound is all select A needs -> one element in eight
                            tie_hi += acc.popc_mask(__builtin_amdgcn_ballot_w64(k0 == t_hi));
                            tie_lo += acc.popc_mask(__builtin_amdgcn_ballot_w64(k0 == t_lo));
                        }
                        if (umax(umax(d0, d1), umax(d2, d3)) > span) {
                            if (d0 > span) rare(k0);
                            if (d1 > span) rare(k1_);
                            if (d2 > span) rare(k2);
                            if (d3 > span) rare(k3);
                        }
                    }
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
            const uint32_t e0 = k * kQ1TileElems + threadIdx.x;
#pragma unroll 4
            for (int r = 0; r < 4 * kQ1U; r++) {
                const uint32_t i = e0 + r * kQ1Block;
                const bool in = i < n;
                const uint32_t key = f2key(in ? x[i] : 0.f);
                const int b = (int)(key >> 20);
                if ((r & 3) == 0) acc.elect(b, in);
                acc.template commit<false>(b, in);
                if (SPEC && in && key - t_lo > span) rare(key);
            }
        }
        acc.flush_hot();
        acc.hot_bin = -1;
        if (SPEC && spec && (threadIdx.x & 63) == 0) {
            if (tie_hi) atomicAdd(&ties[0], (uint32_t)tie_hi);
            if (tie_lo) atomicAdd(&ties[1], (uint32_t)tie_lo);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the inline-assembly ds_adds are invisible to the compiler
        __syncthreads();
        if (spec) {
            if (threadIdx.x < 2) {                                 // reserve this workgroup's slice of the job's lists
                const uint32_t all = staged_n[threadIdx.x];
                staged_base[threadIdx.x] = all ? atomicAdd(&P[threadIdx.x ? kPCntLo : kPCntHi], min(all, kQ1LocalCap)) : 0u;
                if (all > kQ1LocalCap) P[threadIdx.x ? kPOvfLo : kPOvfHi] = 1u;
            }
            if (threadIdx.x < 2 && ties[threadIdx.x]) atomicAdd(&P[threadIdx.x ? kPTieLo : kPTieHi], ties[threadIdx.x]);
            __syncthreads();
            for (int w = 0; w < 2; w++) {
                const uint32_t cnt = min(staged_n[w], kQ1LocalCap), at = staged_base[w];
                uint32_t* list = job.spec + (w ? job.cap : 0u);
                for (uint32_t i = threadIdx.x; i < cnt; i += kQ1Block)
                    if (at + i < job.cap) list[at + i] = staged[w][i];
            }
        }
        for (int i = threadIdx.x; i < kQ1; i += kQ1Block) {        // flush + zero the LDS histogram copies
            int v = 0;
#pragma unroll
            for (int cpy = 0; cpy < kQ1Copies; cpy++) { v += h[cpy * kQ1 + i]; h[cpy * kQ1 + i] = 0; }
            if (v) atomicAdd(&job.ws[kOffH1 + i], (uint32_t)v);
        }
        __syncthreads();
        if (threadIdx.x < 2) { staged_n[threadIdx.x] = 0; ties[threadIdx.x] = 0; }
        __syncthreads();
    }
}

template <int STEP>
__global__ __launch_bounds__(kBlock) void quantile_multi_kernel(const QuantileJobs jobs) {
    // one workgroup per job (select A: per job and side)
    constexpr bool per_job = STEP == kQSelectA || STEP == kQSelectB || STEP == kQPick || STEP == kQSelect0;
    uint32_t lo = per_job ? (STEP == kQSelectA ? blockIdx.x >> 1 : blockIdx.x) : 0, hi = jobs.count;
    while (!per_job && hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs.job[mid].first_block <= blockIdx.x) lo = mid; else hi = mid;
    }
    const QuantileJob& j = jobs.job[lo];
    QuantileCtx c;
    c.x = j.x; c.ws = j.ws; c.spec = j.spec; c.dest = j.dest; c.n = j.n; c.k_hi = j.k_hi; c.k_lo = j.k_lo; c.cap = j.cap;
    const uint32_t end = lo + 1 < jobs.count ? jobs.job[lo + 1].first_block : gridDim.x;
    const uint32_t bidx = blockIdx.x - j.first_block, nblk = end - j.first_block;
    if (STEP == kQSample) quantile_sample_body(c, bidx, nblk);
    if (STEP == kQSelect0) quantile_select0_body(c);
    if (STEP == kQSelectA) quantile_select_a_body(c, (int)(blockIdx.x & 1u));
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
    const int64_t q = ((int64_t)kQWords + 2 * (int64_t)quantile_spec_cap((uint64_t)(n > 0 ? n : 0))) * 4;
    const int64_t iso = (int64_t)kIsotoneBlocks * (int64_t)sizeof(Top2);
    return q > iso ? q : iso;
}

static int quantile_multi_impl(const ppqhip_quantile_job* jobs, int num_jobs, float q, void* workspace, hipStream_t s,
                               const char* what) {
    uint32_t* ws = (uint32_t*)workspace;
    if (int st = check_hip(hipMemsetAsync(ws, 0, (size_t)num_jobs * kQWords * 4, s), "memset quantile workspace"))
        return st;
    uint32_t* spec_at = ws + (size_t)num_jobs * kQWords;     // the speculative lists live behind all fixed parts
    for (int base = 0; base < num_jobs; base += kQuantileMultiMax) {
        QuantileJobs args;
        args.count = (uint32_t)((num_jobs - base) < kQuantileMultiMax ? (num_jobs - base) : kQuantileMultiMax);
        uint32_t blocks = 0, tiles = 0;
        int64_t elems = 0;
        for (uint32_t k = 0; k < args.count; k++) {
            const ppqhip_quantile_job& src = jobs[base + k];
            const int64_t n = src.n;
            elems += n;
            // index rule of _Quantile_T, sort.cu:13-19: __float2int_rn(num_of_elements * q), clipped to [0, n-1]
            auto pos = [n](float f) -> uint32_t {
                float p = nearbyintf((float)n * f);
                if (!(p > 0.f)) return 0u;                      // also NaN
                if (p >= (float)(n - 1)) return (uint32_t)(n - 1);
                return (uint32_t)p;
            };
            QuantileJob& d = args.job[k];
            d.x = src.x; d.dest = src.dest; d.n = (uint32_t)n; d.ws = ws + (size_t)(base + k) * kQWords;
            d.cap = quantile_spec_cap((uint64_t)n); d.spec = spec_at; spec_at += 2 * (size_t)d.cap;
            d.k_hi = pos(q); d.k_lo = pos(1 - q); d.first_block = blocks;
            d.first_tile = tiles;
            tiles += q1_job_tiles(d.n, aligned16(src.x));
            uint32_t nb = (uint32_t)((n + kQuantileMultiChunk - 1) / kQuantileMultiChunk);
            if (nb > kQuantileMultiCap) nb = kQuantileMultiCap;
            if (nb < 1) nb = 1;
            blocks += nb;
        }
        args.total_tiles = tiles;
        uint32_t g1 = tiles / 2;                  // persistent pass 1: >= 2 tiles per workgroup, <= 2 workgroups per CU
        if (g1 < 1) g1 = 1;
        if (g1 > (uint32_t)(kNumCU * kQ1WgPerCu)) g1 = kNumCU * kQ1WgPerCu;
        const dim3 all(blocks), one(args.count), wg(kBlock);
        if (elems >= kQSpeculateMinElems) {       // sample -> thresholds -> pass 1 with speculative lists
            hipLaunchKernelGGL(quantile_multi_kernel<kQSample>, all, wg, 0, s, args);
            hipLaunchKernelGGL(quantile_multi_kernel<kQSelect0>, one, wg, 0, s, args);
            hipLaunchKernelGGL((quantile_pass1_persistent_kernel<true>), dim3(g1), dim3(kQ1Block), 0, s, args);
        } else {
            hipLaunchKernelGGL((quantile_pass1_persistent_kernel<false>), dim3(g1), dim3(kQ1Block), 0, s, args);
        }
        hipLaunchKernelGGL(quantile_multi_kernel<kQSelectA>, dim3(2 * args.count), wg, 0, s, args);
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

int64_t ppqhip_quantile_multi_workspace_bytes(int num_jobs, int64_t total_elems) {
    // fixed part per job + the speculative lists: sum over jobs of 2 * clamp(n / 128, 4096, 2^20) keys
    if (num_jobs <= 0) return 0;
    return ((int64_t)num_jobs * (kQWords + 2 * 4096) + 2 * ((total_elems > 0 ? total_elems : 0) / 128)) * 4;
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
