/* TEST INFRASTRUCTURE ONLY -- C entry points onto the reference's OWN linear fake-quant kernel bodies, forward and
 * LSQ backward.
 *
 * `linear_kernels.inc` is NOT in this repository: oracle/Makefile writes it (git-ignored, oracle/_ref/) by running
 * extract_kernels.awk over $(REF)/ppq/csrc/cuda/linear.cu: the text of `_QuantizeTensor_LT`,
 * `_QuantizeTensorVectorize_LT`, `_QuantizeTensor_LC`, `_QuantizeTensorVectorize_LC` (linear.cu:38-86, 132-186),
 * `_QuantizeTensor_LT_B` (:235-282) and `_QuantizeTensor_LC_B` (:326-380).  `common.cuh` is the reference's header.
 * ref_kernel_host.h runs the bodies thread by thread (its header explains BlockReduceSum).
 *
 * Restated here (the `<<<...>>>` host functions cannot be parsed by a host compiler), each citing its lines:
 *   - kernel choice and grid of QuantizeTensor_LT (linear.cu:110-127) and QuantizeTensor_LC (:210-231);
 *   - grid and `grad_factor` of QuantizeTensor_LT_B (:304-309) and QuantizeTensor_LC_B (:398-411); `rsqrtf` there is
 *     the device intrinsic (<= 2 ulp), here 1 / sqrtf -- a factor on grad_s only, which is compared with a tolerance.
 * Output: oracle/_ref/libref_kernels.so. */
#include "common.cuh"
#include "ref_kernel_host.h"

#include "linear_kernels.inc"

extern "C" {

/* linear.cu:88-130 */
void ref_kernel_fq_linear_t(const float* value, int64_t n, const float* scale, const float* offset, int clip_min,
                            int clip_max, int rounding, float* out) {
    constexpr int32_t TPB = 256, VPT = 4;
    if (n % VPT == 0)
        launch((unsigned)NUM_OF_BLOCK_NOLIMIT(n, VPT * TPB), 1, TPB, [&] {
            _QuantizeTensorVectorize_LT<VPT, TPB>((int32_t)n, value, scale, offset, clip_min, clip_max, rounding, out); });
    else
        launch((unsigned)NUM_OF_BLOCK_NOLIMIT(n, TPB), 1, TPB, [&] {
            _QuantizeTensor_LT<TPB>((int32_t)n, value, scale, offset, clip_min, clip_max, rounding, out); });
}

/* linear.cu:188-233 */
void ref_kernel_fq_linear_c(const float* value, int64_t n, int32_t element_per_channel, int32_t num_of_channel,
                            const float* scale, const float* offset, int clip_min, int clip_max, int rounding, float* out) {
    constexpr int32_t TPB = 256, VPT = 4;
    if (element_per_channel % VPT == 0)
        launch((unsigned)NUM_OF_BLOCK_NOLIMIT(n, VPT * TPB), 1, TPB, [&] {
            _QuantizeTensorVectorize_LC<VPT, TPB>((int32_t)n, element_per_channel, num_of_channel, value, scale, offset,
                                                   clip_min, clip_max, rounding, out); });
    else
        launch((unsigned)NUM_OF_BLOCK_NOLIMIT(n, TPB), 1, TPB, [&] {
            _QuantizeTensor_LC<TPB>((int32_t)n, element_per_channel, num_of_channel, value, scale, offset,
                                    clip_min, clip_max, rounding, out); });
}

/* linear.cu:284-324.  partials (optional, n floats): the per-element term each thread hands to the block reduction. */
void ref_kernel_fq_linear_t_bwd(const float* value, const float* grad_y, int64_t n, const float* scale, const float* offset,
                                int clip_min, int clip_max, int rounding, float* grad_x, float* grad_s, float* partials) {
    constexpr int TPB = 1024;
    float grad_factor = 1.0f / sqrtf((float)((double)n * (clip_max - clip_min)));
    grad_s[0] = 0.0f;
    g_partials = partials;
    launch((unsigned)NUM_OF_BLOCK_NOLIMIT(n, TPB), 1, TPB,
           [&] { _QuantizeTensor_LT_B<TPB>((int)n, value, scale, offset, grad_y, clip_min, clip_max, grad_factor, rounding, grad_s, grad_x); },
           [&]() -> int64_t { int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; return i < n ? i : -1; });
    g_partials = nullptr;
}

/* linear.cu:383-433.  partials (optional, num_of_channel x element_per_channel floats): each thread's sum over the
 * outer index (one term per element when the channel axis is the outermost, as for weights). */
void ref_kernel_fq_linear_c_bwd(const float* value, const float* grad_y, int64_t n, int32_t element_per_channel,
                                int32_t num_of_channel, const float* scale, const float* offset, int clip_min, int clip_max,
                                int rounding, float* grad_x, float* grad_s, float* partials) {
    constexpr int32_t VPT = 1024;
    float grad_factor = 1.0f / sqrtf((float)((double)n * clip_max));
    for (int c = 0; c < num_of_channel; c++) grad_s[c] = 0.0f;
    g_partials = partials;
    launch((unsigned)num_of_channel, (unsigned)NUM_OF_BLOCK_NOLIMIT(element_per_channel, VPT), VPT,
           [&] { _QuantizeTensor_LC_B<VPT>((int32_t)n, element_per_channel, num_of_channel, value, scale, offset,
                                           const_cast<float*>(grad_y), clip_min, clip_max, grad_factor, rounding, grad_s, grad_x); },
           [&]() -> int64_t { int64_t e = (int64_t)blockIdx.y * VPT + threadIdx.x;
                              return e < element_per_channel ? (int64_t)blockIdx.x * element_per_channel + e : -1; });
    g_partials = nullptr;
}

}  // extern "C"
