/* TEST INFRASTRUCTURE ONLY -- C entry points onto the reference's OWN floating (FP8) fake-quant kernel bodies.
 *
 * `floating_kernels.inc` is NOT in this repository: oracle/Makefile writes it (git-ignored, oracle/_ref/) by running
 * extract_kernels.awk over $(REF)/ppq/csrc/cuda/floating.cu: the text of `_QuantizeTensor_FT` (floating.cu:36-55),
 * `_QuantizeTensor_FC` (:74-97), `_QuantizeTensor_FT_B` (:133-182) and `_QuantizeTensor_FC_B` (:223-283).
 * `common.cuh` is the reference's header (QuantizeScalarFloating).  Restated here: the grids of the host launchers
 * (floating.cu:57-72, 99-131, 184-221, 285-336; NUM_OF_BLOCK caps at 2560 blocks, common.cuh:61-63).
 * Output: oracle/_ref/libref_kernels.so. */
#include "common.cuh"
#include "ref_kernel_host.h"

#include "floating_kernels.inc"

extern "C" {

void ref_kernel_fq_float_t(const float* value, int64_t n, const float* scale, const float* offset, int exponent, int mantissa,
                           float clip_min, float clip_max, int rounding, float* out) {
    launch((unsigned)NUM_OF_BLOCK(n, CUDA_NUM_THREADS), 1, (unsigned)CUDA_NUM_THREADS, [&] {
        _QuantizeTensor_FT(n, value, scale, offset, exponent, mantissa, clip_min, clip_max, rounding, out); });
}

void ref_kernel_fq_float_c(const float* value, int64_t n, int64_t element_per_channel, int num_of_channel, const float* scale,
                           const float* offset, int exponent, int mantissa, float clip_min, float clip_max, int rounding,
                           float* out) {
    launch((unsigned)NUM_OF_BLOCK(n, CUDA_NUM_THREADS), 1, (unsigned)CUDA_NUM_THREADS, [&] {
        _QuantizeTensor_FC(n, element_per_channel, num_of_channel, value, scale, offset, exponent, mantissa, clip_min,
                           clip_max, rounding, out); });
}

void ref_kernel_fq_float_t_bwd(const float* value, const float* grad_y, int64_t n, const float* scale, const float* offset,
                               int exponent, int mantissa, float clip_min, float clip_max, int rounding, float* grad_x,
                               float* grad_s) {
    grad_s[0] = 0.0f;
    for (int64_t i = 0; i < n; i++) grad_x[i] = 0.0f;                       /* at::zeros_like, floating.cu:199 */
    launch((unsigned)NUM_OF_BLOCK(n, 1024), 1, 1024, [&] {
        _QuantizeTensor_FT_B(n, value, scale, offset, grad_y, exponent, mantissa, clip_min, clip_max, rounding, grad_s, grad_x); });
}

void ref_kernel_fq_float_c_bwd(const float* value, const float* grad_y, int64_t n, int64_t element_per_channel,
                               int num_of_channel, const float* scale, const float* offset, int exponent, int mantissa,
                               float clip_min, float clip_max, int rounding, float* grad_x, float* grad_s) {
    for (int c = 0; c < num_of_channel; c++) grad_s[c] = 0.0f;
    for (int64_t i = 0; i < n; i++) grad_x[i] = 0.0f;
    launch((unsigned)num_of_channel, (unsigned)NUM_OF_BLOCK(element_per_channel, 1024), 1024, [&] {
        _QuantizeTensor_FC_B(n, element_per_channel, num_of_channel, value, scale, offset, const_cast<float*>(grad_y), exponent,
                             mantissa, clip_min, clip_max, rounding, grad_s, grad_x); });
}

}  // extern "C"
