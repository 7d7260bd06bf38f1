// hist.hip -- calibration histograms for gfx950 (replaces the global-atomic kernels of
// ppq/csrc/cuda/sort.cu:75-218).
//
// Design: the input is streamed once with 16-B loads; every wavefront owns a PRIVATE copy of the
// histogram in LDS (ds_add_u32, no cross-wave contention; `copies` adapts so a workgroup stays
// within 32 KiB of LDS), the copies are merged after a barrier and only NON-ZERO bins are flushed
// with one global atomic each.  Heavily repeated values (ReLU zeros all land in bin 0) would
// serialise the LDS atomic unit, so every wave keeps one "hot bin" in registers (see WaveHist).
// The number of workgroups is bounded (kHistBlocksPerCU per CU) because every workgroup pays a
// flush of up to `bins` global atomics.
//
// Bin rule == reference: b = floor(|x| / hist_scale) [sym] or floor((x - min) / hist_scale)
// [asym] with IEEE division and a saturating float->int conversion (NaN -> bin 0).
#include <cstdlib>

#include "common.hpp"

namespace ppqhip {

constexpr int kMaxLdsBins = 16384;     // 64 KiB of int32 per copy at most
constexpr int kLdsBudgetInts = 8192;   // target: copies * bins <= 8192 ints (32 KiB) per workgroup
constexpr int kHistU = 4;              // float4 loads in flight per lane

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static int hist_blocks_per_cu() { static int v = env_int("PPQHIP_HIST_BLOCKS_PER_CU", 4); return v; }
static int hist_peel() { static int v = env_int("PPQHIP_HIST_HOT", 1); return v; }
static int hist_copies() { static int v = env_int("PPQHIP_HIST_COPIES", 0); return v; }

struct BinRule {
    float a;      // sym: unused; asym: min
    float hs;     // hist_scale
    int bins;
    int clip;     // clip_outliers
    int asym;
};

// branch-free bin rule; `valid` is cleared for values the reference skips (clip_outliers)
template <bool ASYM>
__device__ __forceinline__ int bin_index(float v, const BinRule& r, bool& valid) {
    const float t = ASYM ? (v - r.a) / r.hs : __builtin_fabsf(v) / r.hs;
    int b = f2i_sat(__builtin_floorf(t));
    const int last = r.bins - 1;
    bool out = b > last;
    if (ASYM) out = out || (b < 0);
    valid = valid && !(out && r.clip);
    b = b > last ? last : b;
    if (ASYM) b = b < 0 ? 0 : b;
    return b;
}

__device__ __forceinline__ bool bin_of(float v, const BinRule& r, int* b_out) {
    bool valid = true;
    *b_out = r.asym ? bin_index<true>(v, r, valid) : bin_index<false>(v, r, valid);
    return valid;
}

// Per-wavefront accumulation into the wave's private LDS histogram.
// HOT enables the hot-bin register: a value that many lanes share (ReLU zeros, saturated or
// already-quantised activations) would serialise the LDS atomic unit (a k-way same-address
// ds_add costs ~k cycles).  Each wave therefore keeps ONE wave-uniform "hot bin" in an SGPR and
// every lane counts its hits on that bin in a VGPR instead of touching LDS; the counter is
// flushed with a single wave reduction when the hot bin changes and at the end.  The hot bin is
// re-elected once per trip (16 elements per lane) from the first element of the trip: the bin of
// the first lane becomes hot when at least kHotMin lanes share it.
constexpr int kHotMin = 12;

template <bool HOT>
struct WaveHist {
    int* h;
    int hot_bin;     // wave-uniform
    int hot_cnt;     // per lane

    __device__ __forceinline__ void init(int* hist) { h = hist; hot_bin = -1; hot_cnt = 0; }

    __device__ __forceinline__ void flush_hot() {
        if (!HOT) return;
        int c = hot_cnt;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) c += __shfl_xor(c, m, 64);
        if ((threadIdx.x & 63) == 0 && c != 0 && hot_bin >= 0) atomicAdd(&h[hot_bin], c);
        hot_cnt = 0;
    }

    // all lanes of the wave must call this together (uniform control flow)
    __device__ __forceinline__ void elect(int b, bool valid) {
        if (!HOT) return;
        const unsigned long long act = __ballot(valid);
        if (act == 0ull) return;
        const int leader = __ffsll((long long)act) - 1;
        const int cand = __builtin_amdgcn_readlane(b, leader);
        if (cand == hot_bin) return;
        const int cnt = __popcll(__ballot(valid && b == cand));
        if (cnt >= kHotMin) { flush_hot(); hot_bin = cand; }
    }

    __device__ __forceinline__ void add(int b, bool valid) {
        if (HOT) {
            const bool is_hot = b == hot_bin;
            hot_cnt += (valid && is_hot) ? 1 : 0;
            valid = valid && !is_hot;
        }
        if (valid) atomicAdd(&h[b], 1);
    }
};

__device__ __forceinline__ void lds_hist_zero(int* lds, int total) {
    for (int i = threadIdx.x; i < total; i += blockDim.x) lds[i] = 0;
    __syncthreads();
}

// partial == nullptr: merge the copies and flush the non-zero bins with global atomics (few
// workgroups).  Otherwise store this workgroup's merged histogram to partial[blockIdx.x][bins] with
// plain coalesced stores; hist_reduce_kernel adds the column sums into the caller's histogram.
// (Hundreds of workgroups x thousands of bins of device-scope atomics on a few KiB of addresses
// serialise in the L2 atomic units and would cost more than the streaming pass itself.)
__device__ __forceinline__ void lds_hist_flush(const int* lds, int bins, int copies, int* __restrict__ hist,
                                               int* __restrict__ partial = nullptr) {
    __syncthreads();
    int* dst = partial ? partial + (size_t)blockIdx.x * bins : nullptr;
    for (int b = threadIdx.x; b < bins; b += blockDim.x) {
        int s = 0;
        for (int c = 0; c < copies; c++) s += lds[c * bins + b];
        if (dst) dst[b] = s;
        else if (s) atomicAdd(&hist[b], s);
    }
}

// column sums of partial[count][bins] into hist.  grid = (ceil(bins / 256), slices): each thread
// sums its bin over one slice of the partial histograms (coalesced 1-KiB rows, 8 loads in flight)
// and issues at most one atomic -- `slices` atomics per bin in total, no long serial chain.
constexpr int kReduceSlices = 16;
__global__ __launch_bounds__(kBlock) void hist_reduce_kernel(const int* __restrict__ partial, int count, int bins,
                                                             int* __restrict__ hist) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= bins) return;
    const int per = (count + gridDim.y - 1) / gridDim.y;
    const int lo = blockIdx.y * per;
    const int hi = min(lo + per, count);
    int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = lo;
    for (; i + 8 <= hi; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] += partial[(size_t)(i + k) * bins + b];
    }
    for (; i < hi; i++) acc[0] += partial[(size_t)i * bins + b];
    const int s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    if (s) atomicAdd(&hist[b], s);
}

template <bool PEEL, bool ASYM>
__global__ __launch_bounds__(kBlock) void hist_t_lds_kernel(const float* __restrict__ x, uint32_t n, int vec_ok,
                                                            BinRule rule, int copies, int* __restrict__ hist,
                                                            int* __restrict__ partial) {
    extern __shared__ int lds[];
    lds_hist_zero(lds, copies * rule.bins);
    WaveHist<PEEL> wh;
    wh.init(lds + ((threadIdx.x >> 6) % copies) * rule.bins);
    const uint32_t stride = gridDim.x * kBlock;
    uint32_t done = 0;
    if (vec_ok) {
        const uint32_t nvec = n >> 2;
        const float4* xv = reinterpret_cast<const float4*>(x);
        // kHistU independent 16-B loads in flight per lane (the loop is otherwise bound by one HBM
        // round trip per trip); uniform trip count so the ballots always see whole wavefronts
        const uint32_t trips = (nvec + stride * kHistU - 1) / (stride * kHistU);
        uint32_t v = blockIdx.x * kBlock + threadIdx.x;
        for (uint32_t t = 0; t < trips; t++, v += stride * kHistU) {
            float4 a[kHistU];
#pragma unroll
            for (int k = 0; k < kHistU; k++)
                a[k] = (v + k * stride < nvec) ? xv[v + k * stride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < kHistU; k++) {
                const bool in = v + k * stride < nvec;
                bool o0 = in, o1 = in, o2 = in, o3 = in;
                const int b0 = bin_index<ASYM>(a[k].x, rule, o0);
                const int b1 = bin_index<ASYM>(a[k].y, rule, o1);
                const int b2 = bin_index<ASYM>(a[k].z, rule, o2);
                const int b3 = bin_index<ASYM>(a[k].w, rule, o3);
                if (k == 0) wh.elect(b0, o0);
                wh.add(b0, o0);
                wh.add(b1, o1);
                wh.add(b2, o2);
                wh.add(b3, o3);
            }
        }
        done = nvec << 2;
    }
    {   // scalar remainder (whole tensor when unaligned)
        const uint32_t rem = n - done;
        const uint32_t trips = (rem + stride - 1) / stride;
        uint32_t i = blockIdx.x * kBlock + threadIdx.x;
        for (uint32_t t = 0; t < trips; t++, i += stride) {
            bool ok = i < rem;
            const int b = bin_index<ASYM>(ok ? x[done + i] : 0.f, rule, ok);
            wh.add(b, ok);
        }
    }
    wh.flush_hot();
    lds_hist_flush(lds, rule.bins, copies, hist, partial);
}

// histograms too large for LDS: global atomics (the reference's strategy)
__global__ __launch_bounds__(kBlock) void hist_t_global_kernel(const float* __restrict__ x, uint32_t n, BinRule rule,
                                                               int* __restrict__ hist) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        int b = 0;
        if (bin_of(x[i], rule, &b)) atomicAdd(&hist[b], 1);
    }
}

// per channel, long rows: workgroup = (row, chunk); one channel per workgroup -> LDS histogram
template <bool PEEL>
__global__ __launch_bounds__(kBlock) void hist_c_row_kernel(const float* __restrict__ x, uint32_t epc, FastDiv chunks,
                                                            FastDiv num_channel, uint32_t chunk_elems, BinRule rule,
                                                            int copies, int* __restrict__ hist) {
    extern __shared__ int lds[];
    lds_hist_zero(lds, copies * rule.bins);
    WaveHist<PEEL> wh;
    wh.init(lds + ((threadIdx.x >> 6) % copies) * rule.bins);
    const uint32_t row = fdiv(blockIdx.x, chunks);
    const uint32_t chunk = blockIdx.x - row * chunks.d;
    const uint32_t c = row - fdiv(row, num_channel) * num_channel.d;
    const uint32_t lo = chunk * chunk_elems;
    const uint32_t hi = min(lo + chunk_elems, epc);
    const float* xr = x + (size_t)row * epc;
    const uint32_t trips = (hi - lo + kBlock - 1) / kBlock;
    uint32_t j = lo + threadIdx.x;
    for (uint32_t t = 0; t < trips; t++, j += kBlock) {
        const bool in = j < hi;
        int b = 0;
        const bool ok = in && bin_of(in ? xr[j] : 0.f, rule, &b);
        if ((t & 15u) == 0) wh.elect(b, ok);
        wh.add(b, ok);
    }
    wh.flush_hot();
    lds_hist_flush(lds, rule.bins, copies, hist + (size_t)c * rule.bins);
}

__global__ __launch_bounds__(kBlock) void hist_c_global_kernel(const float* __restrict__ x, uint32_t n,
                                                               FastDiv elem_per_channel, FastDiv num_channel,
                                                               BinRule rule, int* __restrict__ hist) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        int b = 0;
        if (bin_of(x[i], rule, &b)) {
            const uint32_t row = fdiv(i, elem_per_channel);
            const uint32_t c = row - fdiv(row, num_channel) * num_channel.d;
            atomicAdd(&hist[(size_t)c * rule.bins + b], 1);
        }
    }
}

// fused: out = fake_quant(x) (== fq_linear_t) and hist += histogram(x) (== hist_sym_t), one read
template <int R, bool PEEL>
__global__ __launch_bounds__(kBlock) void fq_linear_t_hist_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    float* __restrict__ out, uint32_t n, int vec_ok, int qmin, int qmax, int rounding, BinRule rule, int copies,
    int* __restrict__ hist, int* __restrict__ partial) {
    extern __shared__ int lds[];
    lds_hist_zero(lds, copies * rule.bins);
    WaveHist<PEEL> wh;
    wh.init(lds + ((threadIdx.x >> 6) % copies) * rule.bins);
    const float s = scale[0];
    const int o = round_offset(offset[0]);
    const uint32_t stride = gridDim.x * kBlock;
    uint32_t done = 0;
    if (vec_ok) {
        const uint32_t nvec = n >> 2;
        const float4* xv = reinterpret_cast<const float4*>(x);
        float4* ov = reinterpret_cast<float4*>(out);
        const uint32_t trips = (nvec + stride * kHistU - 1) / (stride * kHistU);
        uint32_t v = blockIdx.x * kBlock + threadIdx.x;
        for (uint32_t t = 0; t < trips; t++, v += stride * kHistU) {
            float4 a[kHistU];
#pragma unroll
            for (int k = 0; k < kHistU; k++)
                a[k] = (v + k * stride < nvec) ? xv[v + k * stride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < kHistU; k++) {
                const bool in = v + k * stride < nvec;
                if (in) {
                    float4 r;
                    r.x = fq_linear_scalar<R>(a[k].x, s, o, qmin, qmax, rounding);
                    r.y = fq_linear_scalar<R>(a[k].y, s, o, qmin, qmax, rounding);
                    r.z = fq_linear_scalar<R>(a[k].z, s, o, qmin, qmax, rounding);
                    r.w = fq_linear_scalar<R>(a[k].w, s, o, qmin, qmax, rounding);
                    ov[v + k * stride] = r;
                }
                bool o0 = in, o1 = in, o2 = in, o3 = in;
                const int b0 = bin_index<false>(a[k].x, rule, o0);
                const int b1 = bin_index<false>(a[k].y, rule, o1);
                const int b2 = bin_index<false>(a[k].z, rule, o2);
                const int b3 = bin_index<false>(a[k].w, rule, o3);
                if (k == 0) wh.elect(b0, o0);
                wh.add(b0, o0);
                wh.add(b1, o1);
                wh.add(b2, o2);
                wh.add(b3, o3);
            }
        }
        done = nvec << 2;
    }
    {
        const uint32_t rem = n - done;
        const uint32_t trips = (rem + stride - 1) / stride;
        uint32_t i = blockIdx.x * kBlock + threadIdx.x;
        for (uint32_t t = 0; t < trips; t++, i += stride) {
            const bool in = i < rem;
            const float a = in ? x[done + i] : 0.f;
            if (in) out[done + i] = fq_linear_scalar<R>(a, s, o, qmin, qmax, rounding);
            bool ok = in;
            const int b = bin_index<false>(a, rule, ok);
            wh.add(b, ok);
        }
    }
    wh.flush_hot();
    lds_hist_flush(lds, rule.bins, copies, hist, partial);
}

static int validate(int64_t n, int64_t bins, const char* what) {
    if (n <= 0) { set_error("%s: tensor is empty", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (n > 0x7fffffffLL) { set_error("%s: too many elements", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (bins <= 0 || bins > 0x3fffffffLL) {
        set_error("%s: histogram is empty or too large", what); return PPQHIP_ERR_INVALID_VALUE;
    }
    return PPQHIP_OK;
}

static int pick_copies(int bins) {
    int c = kLdsBudgetInts / bins;
    if (hist_copies() > 0) c = hist_copies();
    if (c < 1) c = 1;
    if (c > kBlock / kWave) c = kBlock / kWave;
    return c;
}

static int hist_grid(int64_t n) {
    // >= 4096 elements per workgroup so the flush (<= bins atomics) stays a minor term
    return stream_grid(n, 4096, kNumCU * hist_blocks_per_cu());
}

constexpr int kAtomicFlushMaxBlocks = 8;

// scratch for the two-stage flush, or nullptr when the launch is small enough for atomics
static int* partial_for(int grid, int bins, hipStream_t s, bool* failed) {
    *failed = false;
    if (grid <= kAtomicFlushMaxBlocks) return nullptr;
    int* p = (int*)scratch(s, sizeof(int) * (size_t)grid * bins);
    if (p == nullptr) *failed = true;
    return p;
}

static int launch_hist_t(const float* x, int64_t n, BinRule rule, int32_t* hist, hipStream_t s) {
    if (rule.bins <= kMaxLdsBins) {
        const int copies = pick_copies(rule.bins);
        const size_t lds = sizeof(int) * (size_t)copies * rule.bins;
        const int vec_ok = aligned16(x) ? 1 : 0;
        const int grid = hist_grid(n);
        bool failed;
        int* partial = partial_for(grid, rule.bins, s, &failed);
        if (failed) return PPQHIP_ERR_HIP;
#define PPQ_LAUNCH_HIST(P, A)                                                                                  \
    hipLaunchKernelGGL((hist_t_lds_kernel<P, A>), dim3(grid), dim3(kBlock), lds, s, x, (uint32_t)n, vec_ok, rule, \
                       copies, hist, partial)
        if (hist_peel()) { if (rule.asym) PPQ_LAUNCH_HIST(true, true); else PPQ_LAUNCH_HIST(true, false); }
        else { if (rule.asym) PPQ_LAUNCH_HIST(false, true); else PPQ_LAUNCH_HIST(false, false); }
#undef PPQ_LAUNCH_HIST
        if (partial)
            hipLaunchKernelGGL(hist_reduce_kernel, dim3((rule.bins + kBlock - 1) / kBlock, kReduceSlices), dim3(kBlock), 0, s,
                               (const int*)partial, grid, rule.bins, hist);
    } else {
        hipLaunchKernelGGL(hist_t_global_kernel, dim3(stream_grid(n, kBlock * 4)), dim3(kBlock), 0, s, x, (uint32_t)n,
                           rule, hist);
    }
    return PPQHIP_OK;
}

}  // namespace ppqhip

using namespace ppqhip;

extern "C" {

int ppqhip_hist_sym_t(const float* x, int64_t n, float hist_scale, int clip_outliers, int32_t* hist,
                      int64_t num_bins, void* stream) {
    if (int st = validate(n, num_bins, "hist_sym_t")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_HIST_SYM_T, 4.0 * (double)n, s);
    BinRule rule{0.f, hist_scale, (int)num_bins, clip_outliers ? 1 : 0, 0};
    if (int st = launch_hist_t(x, n, rule, hist, s)) return st;
    return finish_launch("hist_sym_t");
}

int ppqhip_hist_asym_t(const float* x, int64_t n, float min_value, float max_value, int clip_outliers,
                       int32_t* hist, int64_t num_bins, void* stream) {
    if (int st = validate(n, num_bins, "hist_asym_t")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_HIST_ASYM_T, 4.0 * (double)n, s);
    // float hist_scale = (max - min) / num_of_bins: sort.cu:123 (float / int64 -> float)
    const float hs = (max_value - min_value) / (float)num_bins;
    BinRule rule{min_value, hs, (int)num_bins, clip_outliers ? 1 : 0, 1};
    if (int st = launch_hist_t(x, n, rule, hist, s)) return st;
    return finish_launch("hist_asym_t");
}

int ppqhip_hist_sym_c(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                      float hist_scale, int clip_outliers, int32_t* hist, int64_t num_bins, void* stream) {
    if (int st = validate(n, num_bins, "hist_sym_c")) return st;
    if (num_channel <= 0 || elem_per_channel <= 0 || n % (num_channel * elem_per_channel) != 0) {
        set_error("hist_sym_c: Kernel Failure, Histogram shape is invalid."); return PPQHIP_ERR_INVALID_VALUE;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_HIST_SYM_C, 4.0 * (double)n, s);
    BinRule rule{0.f, hist_scale, (int)num_bins, clip_outliers ? 1 : 0, 0};
    const FastDiv nc = make_fastdiv((uint32_t)num_channel);
    if (elem_per_channel >= 1024 && num_bins <= kMaxLdsBins) {
        const int copies = pick_copies(rule.bins);
        const size_t lds = sizeof(int) * (size_t)copies * rule.bins;
        const uint32_t chunk_elems = 16384;
        const uint32_t chunks = (uint32_t)((elem_per_channel + chunk_elems - 1) / chunk_elems);
        const int64_t rows = n / elem_per_channel;
        hipLaunchKernelGGL((hist_c_row_kernel<true>), dim3((uint32_t)(rows * chunks)), dim3(kBlock), lds, s, x,
                           (uint32_t)elem_per_channel, make_fastdiv(chunks), nc, chunk_elems, rule, copies, hist);
    } else {
        hipLaunchKernelGGL(hist_c_global_kernel, dim3(stream_grid(n, kBlock * 4)), dim3(kBlock), 0, s, x, (uint32_t)n,
                           make_fastdiv((uint32_t)elem_per_channel), nc, rule, hist);
    }
    return finish_launch("hist_sym_c");
}

int ppqhip_fq_linear_t_hist_sym(const float* x, const float* scale, const float* offset, float* out, int64_t n,
                                int clip_min, int clip_max, int rounding, float hist_scale, int clip_outliers,
                                int32_t* hist, int64_t num_bins, void* stream) {
    if (int st = validate(n, num_bins, "fq_linear_t_hist_sym")) return st;
    if (num_bins > kMaxLdsBins) {
        set_error("fq_linear_t_hist_sym: at most %d bins", kMaxLdsBins); return PPQHIP_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_FQ_HIST_FUSED, 8.0 * (double)n, s);
    BinRule rule{0.f, hist_scale, (int)num_bins, clip_outliers ? 1 : 0, 0};
    const int copies = pick_copies(rule.bins);
    const size_t lds = sizeof(int) * (size_t)copies * rule.bins;
    const int vec_ok = (aligned16(x) && aligned16(out)) ? 1 : 0;
    const int grid = hist_grid(n);
    bool failed;
    int* partial = partial_for(grid, rule.bins, s, &failed);
    if (failed) return PPQHIP_ERR_HIP;
#define PPQ_LAUNCH_FUSED(R, H)                                                                                      \
    hipLaunchKernelGGL((fq_linear_t_hist_kernel<R, H>), dim3(grid), dim3(kBlock), lds, s, x, scale, offset, out,    \
                       (uint32_t)n, vec_ok, clip_min, clip_max, rounding, rule, copies, hist, partial)
    if (rounding == ROUND_HALF_EVEN) { if (hist_peel()) PPQ_LAUNCH_FUSED(ROUND_HALF_EVEN, true); else PPQ_LAUNCH_FUSED(ROUND_HALF_EVEN, false); }
    else { if (hist_peel()) PPQ_LAUNCH_FUSED(-1, true); else PPQ_LAUNCH_FUSED(-1, false); }
#undef PPQ_LAUNCH_FUSED
    if (partial)
        hipLaunchKernelGGL(hist_reduce_kernel, dim3((rule.bins + kBlock - 1) / kBlock, kReduceSlices), dim3(kBlock), 0, s,
                           (const int*)partial, grid, rule.bins, hist);
    return finish_launch("fq_linear_t_hist_sym");
}

}  // extern "C"
