// reduce.hip -- range reductions and order statistics for gfx950.
//
//   minmax_t / minmax_c  one streaming pass (16-B loads), wave64 shuffle reduction, LDS across the
//                        4 waves of a workgroup, ONE pair of global atomics per workgroup on plain
//                        float storage (sign-split integer min/max).  Replaces the
//                        transpose+flatten+torch.min/max of TorchMinMaxObserver.observe
//                        (ppq/quantization/observer/range.py:86-98).
//   quantile_t           replaces Quantile_T (ppq/csrc/cuda/sort.cu:42-59): instead of a full
//                        thrust::sort of a clone it radix-SELECTS the two order statistics on the
//                        order-preserving uint32 key of the floats in three streaming passes
//                        (12 + 12 + 8 bits, LDS histograms), no data movement.
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

// many tensors, one launch (see hist_t_multi_kernel): job j owns workgroups [first_block[j],
// first_block[j+1]); workgroup b of a job folds its chunk into slot b of that job's persistent slots.
constexpr int kMinMaxMultiMax = 96;              // jobs per launch (2.3 KB of kernel arguments)
constexpr uint32_t kMinMaxMultiChunk = 64u << 10;   // elements per workgroup (256 KB)
struct MinMaxJob {
    const float* x;
    float* slots;
    uint32_t n;
    uint32_t first_block;
};
struct MinMaxJobs {
    MinMaxJob job[kMinMaxMultiMax];
    uint32_t count;
};

__global__ __launch_bounds__(kBlock) void minmax_t_multi_kernel(const MinMaxJobs jobs) {
    __shared__ float lds[16];
    uint32_t lo = 0, hi = jobs.count;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs.job[mid].first_block <= blockIdx.x) lo = mid; else hi = mid;
    }
    const MinMaxJob& j = jobs.job[lo];
    const uint32_t end = lo + 1 < jobs.count ? jobs.job[lo + 1].first_block : gridDim.x;
    const uint32_t bidx = blockIdx.x - j.first_block, nblk = end - j.first_block;
    float mn = INFINITY, mx = -INFINITY;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(j.x) & 15u) == 0;
    stream_elems<4, false>(j.x, j.n, vec_ok, [&](float a) { mn = fminf(mn, a); mx = fmaxf(mx, a); }, bidx, nblk);
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { lds[wid] = mn; lds[8 + wid] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / kWave; w++) { mn = fminf(mn, lds[w]); mx = fmaxf(mx, lds[8 + w]); }
        j.slots[2 * bidx] = fminf(mn, j.slots[2 * bidx]);
        j.slots[2 * bidx + 1] = fmaxf(mx, j.slots[2 * bidx + 1]);
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

// workspace layout (uint32 words)
constexpr int kQ1 = 4096, kQ2 = 4096, kQ3 = 256;
constexpr int kOffH1 = 0;                   // hist1[4096]        : key >> 20
constexpr int kOffH2 = kOffH1 + kQ1;        // hist2[2][4096]     : (key >> 8) & 0xFFF | prefix12 match
constexpr int kOffH3 = kOffH2 + 2 * kQ2;    // hist3[2][256]      : key & 0xFF        | prefix24 match
constexpr int kQWords = kOffH3 + 2 * kQ3;

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

__device__ __forceinline__ void quantile_pass1_body(const float* __restrict__ x, uint32_t n, bool vec_ok,
                                                    uint32_t* __restrict__ ws, uint32_t bidx, uint32_t nblk) {
    __shared__ uint32_t h[kQ1 + kQTrash];
    for (int i = threadIdx.x; i < kQ1; i += kBlock) h[i] = 0;
    __syncthreads();
    HotCounter hc;
    hc.init(h, kQ1);
    stream_tiles<4>(x, n, vec_ok,
                    [&](float v, bool in) { hc.elect((int)(f2key(v) >> 20), in); },
                    [&](float v, bool in) { hc.add(in ? (int)(f2key(v) >> 20) : hc.trash()); }, bidx, nblk);
    hc.flush();
    __syncthreads();
    for (int i = threadIdx.x; i < kQ1; i += kBlock)
        if (h[i]) atomicAdd(&ws[kOffH1 + i], h[i]);
}

__global__ __launch_bounds__(kBlock) void quantile_pass1_kernel(const float* __restrict__ x, uint32_t n, bool vec_ok,
                                                                uint32_t* __restrict__ ws) {
    quantile_pass1_body(x, n, vec_ok, ws, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void quantile_pass2_body(const float* __restrict__ x, uint32_t n, bool vec_ok,
                                                    uint32_t k_hi, uint32_t k_lo, uint32_t* __restrict__ ws,
                                                    uint32_t bidx, uint32_t nblk) {
    __shared__ uint32_t h[2 * (kQ2 + kQTrash)];
    __shared__ uint32_t scratch[kBlock];
    __shared__ uint32_t sel[2];
    select_bin(ws + kOffH1, kQ1, k_hi, scratch, sel);
    const uint32_t p_hi = sel[0];
    __syncthreads();
    select_bin(ws + kOffH1, kQ1, k_lo, scratch, sel);
    const uint32_t p_lo = sel[0];
    for (int i = threadIdx.x; i < 2 * (kQ2 + kQTrash); i += kBlock) h[i] = 0;
    __syncthreads();
    HotCounter hi_c, lo_c;
    hi_c.init(h, kQ2);
    lo_c.init(h + kQ2 + kQTrash, kQ2);
    stream_tiles<4>(x, n, vec_ok,
                    [&](float v, bool in) {
                        const uint32_t key = f2key(v);
                        hi_c.elect((int)((key >> 8) & 0xFFFu), in && (key >> 20) == p_hi);
                        lo_c.elect((int)((key >> 8) & 0xFFFu), in && (key >> 20) == p_lo);
                    },
                    [&](float v, bool in) {
                        const uint32_t key = f2key(v);
                        const uint32_t top = key >> 20;
                        const int mid = (int)((key >> 8) & 0xFFFu);
                        if (in && top == p_hi) hi_c.add(mid);      // rare unless the bin is hot (then it is
                        if (in && top == p_lo) lo_c.add(mid);      // absorbed by the hot register)
                    }, bidx, nblk);
    hi_c.flush(); lo_c.flush();
    __syncthreads();
    for (int i = threadIdx.x; i < kQ2; i += kBlock) {
        if (h[i]) atomicAdd(&ws[kOffH2 + i], h[i]);
        if (h[kQ2 + kQTrash + i]) atomicAdd(&ws[kOffH2 + kQ2 + i], h[kQ2 + kQTrash + i]);
    }
}

__global__ __launch_bounds__(kBlock) void quantile_pass2_kernel(const float* __restrict__ x, uint32_t n, bool vec_ok,
                                                                uint32_t k_hi, uint32_t k_lo,
                                                                uint32_t* __restrict__ ws) {
    quantile_pass2_body(x, n, vec_ok, k_hi, k_lo, ws, blockIdx.x, gridDim.x);
}

// shared by pass 3 and the final pick: 24-bit prefixes and residual ranks of both targets
__device__ void select_prefix24(const uint32_t* __restrict__ ws, uint32_t k_hi, uint32_t k_lo, uint32_t* scratch,
                                uint32_t* sel, uint32_t* p24, uint32_t* r24) {
    const uint32_t ks[2] = {k_hi, k_lo};
    for (int w = 0; w < 2; w++) {
        select_bin(ws + kOffH1, kQ1, ks[w], scratch, sel);
        const uint32_t top = sel[0], r1 = sel[1];
        __syncthreads();
        select_bin(ws + kOffH2 + w * kQ2, kQ2, r1, scratch, sel);
        p24[w] = (top << 12) | sel[0];
        r24[w] = sel[1];
        __syncthreads();
    }
}

__device__ __forceinline__ void quantile_pass3_body(const float* __restrict__ x, uint32_t n, bool vec_ok,
                                                    uint32_t k_hi, uint32_t k_lo, uint32_t* __restrict__ ws,
                                                    uint32_t bidx, uint32_t nblk) {
    __shared__ uint32_t h[2 * (kQ3 + kQTrash)];
    __shared__ uint32_t scratch[kBlock];
    __shared__ uint32_t sel[2];
    uint32_t p24[2], r24[2];
    select_prefix24(ws, k_hi, k_lo, scratch, sel, p24, r24);
    for (int i = threadIdx.x; i < 2 * (kQ3 + kQTrash); i += kBlock) h[i] = 0;
    __syncthreads();
    HotCounter hi_c, lo_c;
    hi_c.init(h, kQ3);
    lo_c.init(h + kQ3 + kQTrash, kQ3);
    stream_tiles<4>(x, n, vec_ok,
                    [&](float v, bool in) {
                        const uint32_t key = f2key(v);
                        hi_c.elect((int)(key & 0xFFu), in && (key >> 8) == p24[0]);
                        lo_c.elect((int)(key & 0xFFu), in && (key >> 8) == p24[1]);
                    },
                    [&](float v, bool in) {
                        const uint32_t key = f2key(v);
                        const int low = (int)(key & 0xFFu);
                        if (in && (key >> 8) == p24[0]) hi_c.add(low);
                        if (in && (key >> 8) == p24[1]) lo_c.add(low);
                    }, bidx, nblk);
    hi_c.flush(); lo_c.flush();
    __syncthreads();
    for (int i = threadIdx.x; i < kQ3; i += kBlock) {
        if (h[i]) atomicAdd(&ws[kOffH3 + i], h[i]);
        if (h[kQ3 + kQTrash + i]) atomicAdd(&ws[kOffH3 + kQ3 + i], h[kQ3 + kQTrash + i]);
    }
}

__global__ __launch_bounds__(kBlock) void quantile_pass3_kernel(const float* __restrict__ x, uint32_t n, bool vec_ok,
                                                                uint32_t k_hi, uint32_t k_lo,
                                                                uint32_t* __restrict__ ws) {
    quantile_pass3_body(x, n, vec_ok, k_hi, k_lo, ws, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void quantile_pick_body(uint32_t k_hi, uint32_t k_lo, const uint32_t* __restrict__ ws,
                                                   float* __restrict__ dest) {
    __shared__ uint32_t scratch[kBlock];
    __shared__ uint32_t sel[2];
    uint32_t p24[2], r24[2];
    select_prefix24(ws, k_hi, k_lo, scratch, sel, p24, r24);
    for (int w = 0; w < 2; w++) {
        select_bin(ws + kOffH3 + w * kQ3, kQ3, r24[w], scratch, sel);
        if (threadIdx.x == 0) dest[w] = key2f((p24[w] << 8) | sel[0]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlock) void quantile_pick_kernel(uint32_t k_hi, uint32_t k_lo,
                                                               const uint32_t* __restrict__ ws,
                                                               float* __restrict__ dest) {
    quantile_pick_body(k_hi, k_lo, ws, dest);
}

// many tensors, one launch per pass (see hist_t_multi_kernel): job j owns workgroups
// [first_block[j], first_block[j+1]) and its own kQWords-word slice of the workspace.
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

template <int PASS>
__global__ __launch_bounds__(kBlock) void quantile_multi_kernel(const QuantileJobs jobs) {
    if (PASS == 4) {                                   // pick: one workgroup per job
        const QuantileJob& j = jobs.job[blockIdx.x];
        quantile_pick_body(j.k_hi, j.k_lo, j.ws, j.dest);
        return;
    }
    uint32_t lo = 0, hi = jobs.count;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs.job[mid].first_block <= blockIdx.x) lo = mid; else hi = mid;
    }
    const QuantileJob& j = jobs.job[lo];
    const uint32_t end = lo + 1 < jobs.count ? jobs.job[lo + 1].first_block : gridDim.x;
    const uint32_t bidx = blockIdx.x - j.first_block, nblk = end - j.first_block;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(j.x) & 15u) == 0;
    if (PASS == 1) quantile_pass1_body(j.x, j.n, vec_ok, j.ws, bidx, nblk);
    if (PASS == 2) quantile_pass2_body(j.x, j.n, vec_ok, j.k_hi, j.k_lo, j.ws, bidx, nblk);
    if (PASS == 3) quantile_pass3_body(j.x, j.n, vec_ok, j.k_hi, j.k_lo, j.ws, bidx, nblk);
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

int ppqhip_minmax_t_slots(const float* x, int64_t n, float* slots, void* stream) {
    if (slots == nullptr) { set_error("minmax_t_slots: slots is null"); return PPQHIP_ERR_INVALID_VALUE; }
    return minmax_t_impl(x, n, nullptr, nullptr, slots, stream);
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
    const uint32_t max_slots = (uint32_t)ppqhip_minmax_slots();
    for (int base = 0; base < num_jobs; base += kMinMaxMultiMax) {
        MinMaxJobs args;
        args.count = (uint32_t)((num_jobs - base) < kMinMaxMultiMax ? (num_jobs - base) : kMinMaxMultiMax);
        uint32_t blocks = 0;
        for (uint32_t k = 0; k < args.count; k++) {
            const ppqhip_minmax_job& src = jobs[base + k];
            args.job[k].x = src.x; args.job[k].slots = src.slots; args.job[k].n = (uint32_t)src.n;
            args.job[k].first_block = blocks;
            uint32_t nb = (uint32_t)((src.n + kMinMaxMultiChunk - 1) / kMinMaxMultiChunk);
            if (nb > max_slots) nb = max_slots;
            if (nb < 1) nb = 1;
            blocks += nb;
        }
        hipLaunchKernelGGL(minmax_t_multi_kernel, dim3(blocks), dim3(kBlock), 0, s, args);
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

int ppqhip_quantile_t(const float* x, int64_t n, float q, float* dest, void* workspace, void* stream) {
    if (int st = validate(n, "quantile_t")) return st;
    if (workspace == nullptr) { set_error("quantile_t: workspace is null"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_QUANTILE, 4.0 * (double)n, s);
    // index rule of _Quantile_T, sort.cu:13-19: __float2int_rn(num_of_elements * q), clipped to [0, n-1]
    auto pos = [n](float f) -> uint32_t {
        float p = nearbyintf((float)n * f);
        if (!(p > 0.f)) return 0u;                      // also NaN
        if (p >= (float)(n - 1)) return (uint32_t)(n - 1);
        return (uint32_t)p;
    };
    const uint32_t k_hi = pos(q), k_lo = pos(1 - q);
    uint32_t* ws = (uint32_t*)workspace;
    if (int st = check_hip(hipMemsetAsync(ws, 0, (size_t)kQWords * 4, s), "memset quantile workspace")) return st;
    static const int qpc = getenv("PPQHIP_Q_PER_CU") ? atoi(getenv("PPQHIP_Q_PER_CU")) : 4;
    const int grid = stream_grid(n, kBlock * 16, kNumCU * qpc);
    const bool vec_ok = aligned16(x);
    hipLaunchKernelGGL(quantile_pass1_kernel, dim3(grid), dim3(kBlock), 0, s, x, (uint32_t)n, vec_ok, ws);
    hipLaunchKernelGGL(quantile_pass2_kernel, dim3(grid), dim3(kBlock), 0, s, x, (uint32_t)n, vec_ok, k_hi, k_lo, ws);
    hipLaunchKernelGGL(quantile_pass3_kernel, dim3(grid), dim3(kBlock), 0, s, x, (uint32_t)n, vec_ok, k_hi, k_lo, ws);
    hipLaunchKernelGGL(quantile_pick_kernel, dim3(1), dim3(kBlock), 0, s, k_hi, k_lo, ws, dest);
    return finish_launch("quantile_t");
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
            auto pos = [n](float f) -> uint32_t {               // sort.cu:13-19, as in ppqhip_quantile_t
                float p = nearbyintf((float)n * f);
                if (!(p > 0.f)) return 0u;
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
        hipLaunchKernelGGL(quantile_multi_kernel<1>, dim3(blocks), dim3(kBlock), 0, s, args);
        hipLaunchKernelGGL(quantile_multi_kernel<2>, dim3(blocks), dim3(kBlock), 0, s, args);
        hipLaunchKernelGGL(quantile_multi_kernel<3>, dim3(blocks), dim3(kBlock), 0, s, args);
        hipLaunchKernelGGL(quantile_multi_kernel<4>, dim3(args.count), dim3(kBlock), 0, s, args);
    }
    return finish_launch("quantile_t_multi");
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
