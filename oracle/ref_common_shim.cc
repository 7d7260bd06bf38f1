/* TEST INFRASTRUCTURE ONLY -- C entry points onto the reference's OWN scalar device functions.
 *
 * `common.cuh` below is NOT in this repository: the include resolves to
 * $(REF)/ppq/csrc/cuda/common.cuh (oracle/Makefile passes -I$(REF)/ppq/csrc/cuda), i.e. the reference's
 * header-only `__device__ __inline__` code compiled as host C++ where it lies, with the stand-in headers
 * of oracle/ref_host_stubs/ in place of cuda.h / cuda_runtime.h / torch/extension.h / ATen/cuda/CUDAContext.h.
 * Output: oracle/_ref/libref_common.so (git-ignored; travels to the GPU box with the other built files).
 *
 * What it pins: `_round2int` (common.cuh:88-114, all 8 modes), `QuantizeScalar` / `DequantizeScalar`
 * (:116-147) and `QuantizeScalarFloating` (:154-226) -- the FP8 function that has no CPU twin and no
 * test in the reference.  The array entry points apply the two statements of the kernel bodies
 * (floating.cu:49-52, 89-94; linear.cu:49-57) around those functions; kernels themselves (`<<<>>>`) do
 * not compile on a host.
 *
 * Host-vs-device caveat: float -> int conversion of a value outside int32 is undefined in host C++
 * (x86 returns INT_MIN) and saturating on the GPU; sweeps that compare against this library keep
 * |value / scale| (and, in the subnormal branch, |value / quantum|) below 2^31. */
#include "common.cuh"

extern "C" {

int ref_round2int(float value, int rounding) { return _round2int(value, rounding); }

int ref_quantize_scalar(float value, float scale, int offset, int clip_min, int clip_max, int rounding) {
    return QuantizeScalar<float, float, int>(value, scale, offset, clip_min, clip_max, rounding);
}

float ref_dequantize_scalar(int value, float scale, int offset) {
    return DequantizeScalar<int, float, int>(value, scale, offset);
}

float ref_quantize_scalar_floating(float value, float scale, float offset, int exponent, int mantissa,
                                   float clip_min, float clip_max, int rounding) {
    return QuantizeScalarFloating<float, float, float>(value, scale, offset, exponent, mantissa, clip_min, clip_max, rounding);
}

/* floating.cu:47-53 (_QuantizeTensor_FT) */
void ref_fq_float_t(const float* value, int64_t n, const float* scale, const float* offset, int exponent, int mantissa,
                    float clip_min, float clip_max, int rounding, float* out) {
    float s = scale[0], o = offset[0];
    for (int64_t i = 0; i < n; i++) {
        float qt = QuantizeScalarFloating<float, float, float>(value[i], s, o, exponent, mantissa, clip_min, clip_max, rounding);
        out[i] = DequantizeScalar<float, float, float>(qt, s, o);
    }
}

/* floating.cu:86-95 (_QuantizeTensor_FC) */
void ref_fq_float_c(const float* value, int64_t n, int64_t element_per_channel, int num_of_channel, const float* scale,
                    const float* offset, int exponent, int mantissa, float clip_min, float clip_max, int rounding, float* out) {
    for (int64_t i = 0; i < n; i++) {
        int c = (i / element_per_channel) % num_of_channel;
        float qt = QuantizeScalarFloating<float, float, float>(value[i], scale[c], offset[c], exponent, mantissa,
                                                                 clip_min, clip_max, rounding);
        out[i] = DequantizeScalar<float, float, float>(qt, scale[c], offset[c]);
    }
}

/* linear.cu:49-57 (_QuantizeTensor_LT): offset rounded half away from zero, then quantize / dequantize */
void ref_fq_linear_t(const float* value, int64_t n, const float* scale, const float* offset, int clip_min, int clip_max,
                     int rounding, float* out) {
    float s = scale[0];
    int o = std::round(offset[0]);
    for (int64_t i = 0; i < n; i++) {
        float qt = QuantizeScalar<float, float, int>(value[i], s, o, clip_min, clip_max, rounding);
        out[i] = DequantizeScalar<int, float, int>(qt, s, o);
    }
}

}  // extern "C"
