// search.hip -- clipping-range searches over calibration histograms, on device and batched over
// many histograms (one launch serves every TensorQuantizationConfig of a graph).
//
//   mse_search   TorchMSEObserver.hist_to_scale_offset's sweep (ppq/quantization/observer/range.py:
//                456-520) with the loss of ppq/csrc/cpu/hist_mse.cc:3-28.  One LANE per candidate
//                (start, step): the lane walks the bins sequentially with the same float operation
//                order as hist_mse.cc, so every loss is bit-identical to the reference's C++ and the
//                first-minimum tie-break of `sorted(...)[0]` (range.py:487, 512) is reproduced by an
//                atomicMin on the packed key (loss_bits << 32 | enumeration index).
//   kl_losses    the per-candidate KL divergences of TorchHistObserver.hist_to_scale_offset
//                (range.py:190-282, measure/statistic.py:3-12): float32 histogram arithmetic, float64
//                divergence.  The arg-min is left to the caller (reference: stable sort, :271).
//   mse_loss_host the reference's host helper itself (CUDA.compute_mse_loss, ffi.py:263-270).
#include <cmath>

#include "common.hpp"

namespace ppqhip {

constexpr int kMseInterval = 8;    // OBSERVER_MSE_COMPUTE_INTERVAL, ppq/core/common.py:28
constexpr int kMseBlock = 256;

// loss of one candidate, hist_mse.cc:9-27 (float accumulation, bins in ascending order)
__device__ __forceinline__ float mse_candidate_loss(const int* __restrict__ h, int bins, float total, int start,
                                                    int step, int end) {
    float loss = 0.0f;
    for (int idx = 0; idx < bins; idx++) {
        float error;
        if (idx < start) error = (float)(start - idx - 1) + 0.5f;
        else if (idx > end) error = (float)(idx - end) + 0.5f;
        else {
            const int l_idx = (idx - start) % step;
            const int r_idx = step - l_idx - 1;
            if (l_idx == r_idx) error = (float)l_idx + 0.25f;
            else {
                const float l_err = (float)l_idx + 0.5f, r_err = (float)r_idx + 0.5f;
                error = l_err < r_err ? l_err : r_err;
            }
        }
        loss += ((float)h[idx] * error * error) / total;
    }
    return loss;
}

// grid = (splits, num_hist).  Candidate cells: cell 0 is the "at least min-max" candidate
// (range.py:472-474); cell 1 + a*S + (step-1) is (start = 8a, step), S = bins // levels.
__global__ __launch_bounds__(kMseBlock) void mse_search_kernel(const int32_t* __restrict__ hist, int bins,
                                                               const double* __restrict__ hist_scale,
                                                               const double* __restrict__ min_value, int levels,
                                                               int symmetrical,
                                                               unsigned long long* __restrict__ best) {
    extern __shared__ int h[];
    const int hid = blockIdx.y;
    const int32_t* src = hist + (size_t)hid * bins;
    for (int i = threadIdx.x; i < bins; i += kMseBlock) h[i] = src[i];
    __syncthreads();
    long long tot = 0;
    for (int i = 0; i < bins; i++) tot += h[i];   // every lane: uniform LDS broadcast reads
    const float total = (float)tot;
    const int S = bins / levels;
    const int n_start = symmetrical ? 1 : (bins + kMseInterval - 1) / kMseInterval;
    const int cells = 1 + n_start * S;
    const double hs = hist_scale[hid], mn = min_value[hid];
    unsigned long long mine = ~0ull;
    for (int cell = blockIdx.x * kMseBlock + threadIdx.x; cell < cells; cell += gridDim.x * kMseBlock) {
        int start, step, end;
        bool valid = true;
        if (cell == 0) { start = 0; step = S + 1; end = levels * step; }
        else {
            const int a = (cell - 1) / S;
            step = (cell - 1) % S + 1;
            start = a * kMseInterval;
            end = start + levels * step;
            // `if (start * hist_scale) + self._min > 0: break` (range.py:477) -- monotone in start
            if (!symmetrical && ((double)start * hs) + mn > 0) valid = false;
            if (end > bins + levels) valid = false;          // range.py:481 / :504, monotone in step
        }
        if (!valid) continue;
        const float loss = mse_candidate_loss(h, bins, total, start, step, end);
        if (loss != loss) continue;
        const unsigned long long key = ((unsigned long long)__float_as_uint(loss) << 32) | (unsigned)cell;
        mine = key < mine ? key : mine;
    }
    if (mine != ~0ull) atomicMin(&best[hid], mine);
}

__global__ void mse_decode_kernel(const unsigned long long* __restrict__ packed, int num_hist, int bins, int levels,
                                  int32_t* __restrict__ best) {
    const int hid = blockIdx.x * blockDim.x + threadIdx.x;
    if (hid >= num_hist) return;
    const int S = bins / levels;
    const unsigned long long key = packed[hid];
    int cell = (int)(key & 0xffffffffu);
    if (key == ~0ull) cell = 0;
    int start, step;
    if (cell == 0) { start = 0; step = S + 1; }
    else { start = ((cell - 1) / S) * kMseInterval; step = (cell - 1) % S + 1; }
    best[hid * 4 + 0] = start;
    best[hid * 4 + 1] = start + levels * step;
    best[hid * 4 + 2] = step;
    best[hid * 4 + 3] = cell;
}

// ---------------------------------------------------------------------------------------- KL
// workgroup = (candidate j, histogram h); bin_range r = (j + 1) * Q, expand ratio e = j + 1.
__device__ __forceinline__ double block_sum_f64(double v, double* lds) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) r += lds[w];
    return r;   // every thread gets the same value (fixed summation order)
}

__global__ __launch_bounds__(kBlock) void kl_losses_kernel(const int32_t* __restrict__ hist, int bins, int Q,
                                                           int ncand, double* __restrict__ losses) {
    extern __shared__ float hf[];          // float(histogram) after the "crucial" edits (range.py:244-245)
    __shared__ double red[kBlock / kWave];
    __shared__ float qgroup[1024];         // per quantisation bin: merged count / #positive bins
    const int j = blockIdx.x, hid = blockIdx.y;
    const int r = (j + 1) * Q, e = j + 1;
    const int32_t* src = hist + (size_t)hid * bins;
    const int z = (int)((double)bins * .002);
    for (int i = threadIdx.x; i < bins; i += kBlock) hf[i] = i < z ? 0.f : (i == z ? 1.f : (float)src[i]);
    __syncthreads();
    // exact integer-valued sums, rounded to float32 once (the reference sums float32 tensors)
    double s_all = 0.0, s_tail = 0.0;
    for (int i = threadIdx.x; i < bins; i += kBlock) { s_all += hf[i]; if (i >= r) s_tail += hf[i]; }
    const float hist_sum = (float)block_sum_f64(s_all, red);
    const float tail = (float)block_sum_f64(s_tail, red);
    // q: merge e bins into one of the Q quantisation bins, spread over its positive bins
    for (int g = threadIdx.x; g < Q; g += kBlock) {
        double gs = 0.0; int cnt = 0;
        for (int k = 0; k < e; k++) { const float v = hf[g * e + k]; gs += v; cnt += v > 0.f; }
        qgroup[g] = (float)gs / (float)(cnt == 0 ? 1 : cnt);
    }
    __syncthreads();
    double qs = 0.0;
    for (int i = threadIdx.x; i < r; i += kBlock) qs += hf[i] > 0.f ? (double)qgroup[i / e] : 0.0;
    const float q_sum = (float)block_sum_f64(qs, red);
    // KL(p || q) = sum p * (log10(p + eps) - log10(q + eps)) in float64 (statistic.py:12)
    double acc = 0.0;
    for (int i = threadIdx.x; i < r; i += kBlock) {
        float p = hf[i];
        if (i == r - 1) p = p + tail;
        p = p / hist_sum;
        const float q = (hf[i] > 0.f ? qgroup[i / e] : 0.f) / q_sum;
        const double pd = (double)p, qd = (double)q;
        acc += pd * (log10(pd + 1e-30) - log10(qd + 1e-30));
    }
    const double kl = block_sum_f64(acc, red);
    if (threadIdx.x == 0) losses[(size_t)hid * ncand + j] = kl;
}

}  // namespace ppqhip

using namespace ppqhip;

extern "C" {

float ppqhip_mse_loss_host(const int64_t* hist, int64_t num_bins, int start, int step, int end) {
    // ppq/csrc/cpu/hist_mse.cc:3-28
    int64_t num_of_elements = 0;
    float loss = 0.0f;
    for (int64_t i = 0; i < num_bins; i++) num_of_elements += hist[i];
    for (int idx = 0; idx < (int)num_bins; idx++) {
        float error;
        const int64_t bin = hist[idx];
        if (idx < start) error = (float)(start - idx - 1) + 0.5f;
        else if (idx > end) error = (float)(idx - end) + 0.5f;
        else {
            const int64_t l_idx = (idx - start) % step;
            const int64_t r_idx = step - l_idx - 1;
            if (l_idx == r_idx) error = (float)l_idx + 0.25f;
            else {
                const float l_err = (float)l_idx + 0.5f, r_err = (float)r_idx + 0.5f;
                error = l_err < r_err ? l_err : r_err;
            }
        }
        loss += ((float)bin * error * error) / (float)num_of_elements;
    }
    return loss;
}

int ppqhip_mse_search(const int32_t* hist, int64_t num_hist, int64_t num_bins, const double* hist_scale,
                      const double* min_value, int quant_min, int quant_max, int symmetrical, int32_t* best,
                      void* workspace, void* stream) {
    const int levels = quant_max - quant_min + 1;
    if (num_hist <= 0 || num_bins <= 0 || num_bins > 16384 || levels <= 0 || num_bins / levels < 1) {
        set_error("mse_search: need 1 <= levels <= bins <= 16384 and at least one histogram");
        return PPQHIP_ERR_INVALID_VALUE;
    }
    if (workspace == nullptr) { set_error("mse_search: workspace is null"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_MSE_SEARCH, 4.0 * (double)num_hist * (double)num_bins, s);
    unsigned long long* packed = (unsigned long long*)workspace;
    if (int st = check_hip(hipMemsetAsync(packed, 0xff, sizeof(unsigned long long) * (size_t)num_hist, s),
                           "memset mse workspace")) return st;
    const int S = (int)(num_bins / levels);
    const int n_start = symmetrical ? 1 : (int)((num_bins + kMseInterval - 1) / kMseInterval);
    const int cells = 1 + n_start * S;
    int splits = (cells + kMseBlock - 1) / kMseBlock;
    if (splits > 64) splits = 64;
    hipLaunchKernelGGL(mse_search_kernel, dim3(splits, (uint32_t)num_hist), dim3(kMseBlock),
                       sizeof(int) * (size_t)num_bins, s, hist, (int)num_bins, hist_scale, min_value, levels,
                       symmetrical ? 1 : 0, packed);
    hipLaunchKernelGGL(mse_decode_kernel, dim3((uint32_t)((num_hist + 63) / 64)), dim3(64), 0, s, packed,
                       (int)num_hist, (int)num_bins, levels, best);
    return finish_launch("mse_search");
}

int64_t ppqhip_mse_search_workspace_bytes(int64_t num_hist) { return 8 * (num_hist > 0 ? num_hist : 1); }

int64_t ppqhip_kl_num_candidates(int64_t num_bins, int num_of_bits) {
    if (num_of_bits < 2 || num_of_bits > 11) return 0;
    const int64_t Q = 1ll << (num_of_bits - 1);
    // range(quant_bins, hist_bins + quant_bins - 1, quant_bins), observer/range.py:248; the
    // reference's reshape only works when bins is a multiple of quant_bins
    if (num_bins < Q || num_bins % Q != 0) return 0;
    return num_bins / Q;
}

int ppqhip_kl_losses(const int32_t* hist, int64_t num_hist, int64_t num_bins, int num_of_bits, double* losses,
                     void* stream) {
    const int64_t ncand = ppqhip_kl_num_candidates(num_bins, num_of_bits);
    if (num_hist <= 0 || ncand <= 0 || num_bins > 16384) {
        set_error("kl_losses: bins must be a multiple of 2^(bits-1), <= 16384, bits in [2, 11]");
        return PPQHIP_ERR_INVALID_VALUE;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_KL_LOSSES, 4.0 * (double)num_hist * (double)num_bins, s);
    const int Q = 1 << (num_of_bits - 1);
    hipLaunchKernelGGL(kl_losses_kernel, dim3((uint32_t)ncand, (uint32_t)num_hist), dim3(kBlock),
                       sizeof(float) * (size_t)num_bins, s, hist, (int)num_bins, Q, (int)ncand, losses);
    return finish_launch("kl_losses");
}

}  // extern "C"
