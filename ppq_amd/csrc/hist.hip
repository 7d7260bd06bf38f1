// hist.hip -- calibration histograms for gfx950 (replaces the global-atomic kernels of
// ppq/csrc/cuda/sort.cu:75-218).
//
// Design: the input is streamed once with 16-B loads; every wavefront owns a PRIVATE copy of the
// histogram in LDS (ds_add_u32, no cross-wave contention; `copies` adapts so a workgroup stays
// within 32 KiB of LDS), the copies are merged after a barrier and only NON-ZERO bins are flushed
// with one global atomic each.  Heavily repeated values (ReLU zeros all land in bin 0) would
// serialise the LDS atomic unit, so each access first peels the bin of the first active lane:
// lanes that share it are counted with one ballot and added by a single lane.
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

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static int hist_blocks_per_cu() { static int v = env_int("PPQHIP_HIST_BLOCKS_PER_CU", 2); return v; }
static int hist_peel() { static int v = env_int("PPQHIP_HIST_PEEL", 1); return v; }

struct BinRule {
    float a;      // sym: unused; asym: min
    float hs;     // hist_scale
    int bins;
    int clip;     // clip_outliers
    int asym;
};

__device__ __forceinline__ bool bin_of(float v, const BinRule& r, int* b_out) {
    int b;
    *b_out = 0;
    if (r.asym) {
        b = f2i_sat(__builtin_floorf((v - r.a) / r.hs));
        if (b > r.bins - 1) { if (r.clip) return false; b = r.bins - 1; }
        if (b < 0) { if (r.clip) return false; b = 0; }
    } else {
        b = f2i_sat(__builtin_floorf(__builtin_fabsf(v) / r.hs));
        if (b > r.bins - 1) { if (r.clip) return false; b = r.bins - 1; }
    }
    *b_out = b;
    return true;
}

// add one observation per active lane into an LDS histogram; PEEL aggregates the hottest bin
template <bool PEEL>
__device__ __forceinline__ void lds_hist_add(int* h, int b, bool valid) {
    if (PEEL) {
        const unsigned long long act = __ballot(valid);
        if (act != 0ull) {
            const int leader = __ffsll((long long)act) - 1;
            const int b0 = __shfl(b, leader, 64);
            const unsigned long long same = __ballot(valid && b == b0);
            const int cnt = __popcll(same);
            if (cnt >= 4) {
                if ((int)(threadIdx.x & 63) == leader) atomicAdd(&h[b0], cnt);
                valid = valid && (b != b0);
            }
        }
    }
    if (valid) atomicAdd(&h[b], 1);
}

__device__ __forceinline__ void lds_hist_zero(int* lds, int total) {
    for (int i = threadIdx.x; i < total; i += blockDim.x) lds[i] = 0;
    __syncthreads();
}

__device__ __forceinline__ void lds_hist_flush(const int* lds, int bins, int copies, int* __restrict__ hist) {
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += blockDim.x) {
        int s = 0;
        for (int c = 0; c < copies; c++) s += lds[c * bins + b];
        if (s) atomicAdd(&hist[b], s);
    }
}

template <bool PEEL>
__global__ __launch_bounds__(kBlock) void hist_t_lds_kernel(const float* __restrict__ x, uint32_t n, int vec_ok,
                                                            BinRule rule, int copies, int* __restrict__ hist) {
    extern __shared__ int lds[];
    lds_hist_zero(lds, copies * rule.bins);
    int* h = lds + ((threadIdx.x >> 6) % copies) * rule.bins;
    const uint32_t stride = gridDim.x * kBlock;
    uint32_t done = 0;
    if (vec_ok) {
        const uint32_t nvec = n >> 2;
        const float4* xv = reinterpret_cast<const float4*>(x);
        // uniform trip count so the ballots in lds_hist_add always see whole wavefronts
        const uint32_t trips = (nvec + stride - 1) / stride;
        uint32_t v = blockIdx.x * kBlock + threadIdx.x;
        for (uint32_t t = 0; t < trips; t++, v += stride) {
            const bool in = v < nvec;
            float4 a = in ? xv[v] : make_float4(0.f, 0.f, 0.f, 0.f);
            int b = 0;
            bool ok;
            ok = in && bin_of(a.x, rule, &b); lds_hist_add<PEEL>(h, b, ok);
            ok = in && bin_of(a.y, rule, &b); lds_hist_add<PEEL>(h, b, ok);
            ok = in && bin_of(a.z, rule, &b); lds_hist_add<PEEL>(h, b, ok);
            ok = in && bin_of(a.w, rule, &b); lds_hist_add<PEEL>(h, b, ok);
        }
        done = nvec << 2;
    }
    {   // scalar remainder (whole tensor when unaligned)
        const uint32_t rem = n - done;
        const uint32_t trips = (rem + stride - 1) / stride;
        uint32_t i = blockIdx.x * kBlock + threadIdx.x;
        for (uint32_t t = 0; t < trips; t++, i += stride) {
            const bool in = i < rem;
            int b = 0;
            const bool ok = in && bin_of(in ? x[done + i] : 0.f, rule, &b);
            lds_hist_add<PEEL>(h, b, ok);
        }
    }
    lds_hist_flush(lds, rule.bins, copies, hist);
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
    int* h = lds + ((threadIdx.x >> 6) % copies) * rule.bins;
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
        lds_hist_add<PEEL>(h, b, ok);
    }
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
    int* __restrict__ hist) {
    extern __shared__ int lds[];
    lds_hist_zero(lds, copies * rule.bins);
    int* h = lds + ((threadIdx.x >> 6) % copies) * rule.bins;
    const float s = scale[0];
    const int o = round_offset(offset[0]);
    const uint32_t stride = gridDim.x * kBlock;
    uint32_t done = 0;
    if (vec_ok) {
        const uint32_t nvec = n >> 2;
        const float4* xv = reinterpret_cast<const float4*>(x);
        float4* ov = reinterpret_cast<float4*>(out);
        const uint32_t trips = (nvec + stride - 1) / stride;
        uint32_t v = blockIdx.x * kBlock + threadIdx.x;
        for (uint32_t t = 0; t < trips; t++, v += stride) {
            const bool in = v < nvec;
            float4 a = in ? xv[v] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (in) {
                float4 r;
                r.x = fq_linear_scalar<R>(a.x, s, o, qmin, qmax, rounding);
                r.y = fq_linear_scalar<R>(a.y, s, o, qmin, qmax, rounding);
                r.z = fq_linear_scalar<R>(a.z, s, o, qmin, qmax, rounding);
                r.w = fq_linear_scalar<R>(a.w, s, o, qmin, qmax, rounding);
                ov[v] = r;
            }
            int b = 0;
            bool ok;
            ok = in && bin_of(a.x, rule, &b); lds_hist_add<PEEL>(h, b, ok);
            ok = in && bin_of(a.y, rule, &b); lds_hist_add<PEEL>(h, b, ok);
            ok = in && bin_of(a.z, rule, &b); lds_hist_add<PEEL>(h, b, ok);
            ok = in && bin_of(a.w, rule, &b); lds_hist_add<PEEL>(h, b, ok);
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
            int b = 0;
            const bool ok = in && bin_of(a, rule, &b);
            lds_hist_add<PEEL>(h, b, ok);
        }
    }
    lds_hist_flush(lds, rule.bins, copies, hist);
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
    if (c < 1) c = 1;
    if (c > kBlock / kWave) c = kBlock / kWave;
    return c;
}

static int hist_grid(int64_t n) {
    // >= 4096 elements per workgroup so the flush (<= bins atomics) stays a minor term
    return stream_grid(n, 4096, kNumCU * hist_blocks_per_cu());
}

static int launch_hist_t(const float* x, int64_t n, BinRule rule, int32_t* hist, hipStream_t s) {
    if (rule.bins <= kMaxLdsBins) {
        const int copies = pick_copies(rule.bins);
        const size_t lds = sizeof(int) * (size_t)copies * rule.bins;
        const int vec_ok = aligned16(x) ? 1 : 0;
        if (hist_peel())
            hipLaunchKernelGGL((hist_t_lds_kernel<true>), dim3(hist_grid(n)), dim3(kBlock), lds, s, x, (uint32_t)n,
                               vec_ok, rule, copies, hist);
        else
            hipLaunchKernelGGL((hist_t_lds_kernel<false>), dim3(hist_grid(n)), dim3(kBlock), lds, s, x, (uint32_t)n,
                               vec_ok, rule, copies, hist);
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
    launch_hist_t(x, n, rule, hist, s);
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
    launch_hist_t(x, n, rule, hist, s);
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
    if (rounding == ROUND_HALF_EVEN)
        hipLaunchKernelGGL((fq_linear_t_hist_kernel<ROUND_HALF_EVEN, true>), dim3(hist_grid(n)), dim3(kBlock), lds, s,
                           x, scale, offset, out, (uint32_t)n, vec_ok, clip_min, clip_max, rounding, rule, copies,
                           hist);
    else
        hipLaunchKernelGGL((fq_linear_t_hist_kernel<-1, true>), dim3(hist_grid(n)), dim3(kBlock), lds, s, x, scale,
                           offset, out, (uint32_t)n, vec_ok, clip_min, clip_max, rounding, rule, copies, hist);
    return finish_launch("fq_linear_t_hist_sym");
}

}  // extern "C"
