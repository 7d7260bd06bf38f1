/*
 * ppq_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the arithmetic of the reference's
 * native kernels (OpenPPL/ppq 0.6.6, ppq/csrc).  It exists so that tests/, smoke() and
 * bench.py's cpu_baseline leg can check / time the HIP product path against the
 * reference's semantics.  NOTHING under ppq_amd/ may import, link or call this file.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/).  The reference kernels are CUDA; three places where CUDA hardware
 * semantics matter are restated explicitly because plain C leaves them undefined:
 *   - float -> int conversion saturates and maps NaN to 0 (cvt.rzi.s32.f32), see sat_f2i;
 *   - `floor(value + .5)` is evaluated in double (the literal .5 is a double);
 *   - the int32 add `round(x/s) + offset` is done in 64 bit and clamped, which equals the
 *     reference whenever the reference's own add does not overflow (signed overflow is UB
 *     there) and equals the reference's PyTorch path (qfunction/linear.py:29-31) otherwise.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* rounding policy values: ppq/csrc/cuda/common.cuh:16-23, ppq/core/quant.py:123-142 */
enum {
    ROUND_HALF_EVEN = 0,
    ROUND_HALF_UP = 1,
    ROUND_HALF_DOWN = 2,
    ROUND_HALF_TOWARDS_ZERO = 3,
    ROUND_HALF_FAR_FORM_ZERO = 4,
    ROUND_TO_NEAR_INT = 5,
    ROUND_UP = 6,
    ROUND_DOWN = 7
};

/* CUDA float/double -> int32 conversion of an already integral value: saturating, NaN -> 0 */
static int32_t sat_d2i(double v) {
    if (v != v) return 0;
    if (v >= 2147483647.0) return INT32_MAX;
    if (v <= -2147483648.0) return INT32_MIN;
    return (int32_t)v;
}
static int32_t sat_f2i(float v) { return sat_d2i((double)v); }

static int32_t clamp_i64(int64_t v, int64_t lo, int64_t hi) {
    if (v > hi) return (int32_t)hi;
    if (v < lo) return (int32_t)lo;
    return (int32_t)v;
}

/* _round2int: ppq/csrc/cuda/common.cuh:88-114 */
int32_t oracle_round2int(float value, int rounding) {
    switch (rounding) {
        case ROUND_HALF_EVEN: return sat_f2i(nearbyintf(value));
        case ROUND_HALF_UP: return sat_d2i(floor((double)value + .5));
        case ROUND_HALF_DOWN: return sat_d2i(ceil((double)value - .5));
        case ROUND_HALF_TOWARDS_ZERO:
            if (value > 0) return oracle_round2int(value, ROUND_HALF_DOWN);
            else return oracle_round2int(value, ROUND_HALF_UP);
        case ROUND_HALF_FAR_FORM_ZERO:
            if (value > 0) return oracle_round2int(value, ROUND_HALF_UP);
            else return oracle_round2int(value, ROUND_HALF_DOWN);
        case ROUND_UP: return sat_f2i(ceilf(value));
        case ROUND_DOWN: return sat_f2i(floorf(value));
        default: return sat_f2i(roundf(value));
    }
}

/* offset rounding used by the linear kernels: `int o = std::round(offset)` linear.cu:52,76,149,175 */
static int32_t round_offset(float o) { return sat_f2i(roundf(o)); }

/* QuantizeScalar + DequantizeScalar: common.cuh:116-147; vector body linear.cu:79-83 */
static float fq_linear_scalar(float x, float s, int32_t o, int qmin, int qmax, int rounding) {
    float qt = x / s;
    int32_t q = clamp_i64((int64_t)oracle_round2int(qt, rounding) + (int64_t)o, qmin, qmax);
    return (float)((int64_t)q - (int64_t)o) * s;
}

/* QuantizeTensor_LT: linear.cu:88-130 (kernels :38-86) */
void oracle_fq_linear_t(const float* x, int64_t n, const float* scale, const float* offset,
                        int qmin, int qmax, int rounding, float* out) {
    float s = scale[0];
    int32_t o = round_offset(offset[0]);
    for (int64_t i = 0; i < n; i++) out[i] = fq_linear_scalar(x[i], s, o, qmin, qmax, rounding);
}

/* QuantizeTensor_LC: linear.cu:188-233 (kernels :132-186); c = (i / epc) % C */
void oracle_fq_linear_c(const float* x, int64_t n, const float* scale, const float* offset,
                        int64_t num_channel, int64_t elem_per_channel,
                        int qmin, int qmax, int rounding, float* out) {
    for (int64_t i = 0; i < n; i++) {
        int64_t c = (i / elem_per_channel) % num_channel;
        out[i] = fq_linear_scalar(x[i], scale[c], round_offset(offset[c]), qmin, qmax, rounding);
    }
}

/* QuantizeTensor_LT_B: linear.cu:235-324.  grad_s is accumulated here in double and
 * multiplied by the reference's float grad_factor; the reference reduces in float over a
 * parallel tree, so grad_s is compared with a tolerance (tests/test_cuda_kernel.py:96-101). */
void oracle_fq_linear_t_bwd(const float* x, const float* dy, int64_t n, const float* scale,
                            const float* offset, int qmin, int qmax, int rounding,
                            float* grad_x, float* grad_s) {
    float o = roundf(offset[0]);
    float s = scale[0];
    float grad_factor = (float)(1.0 / sqrt((double)n * (double)(qmax - qmin))); /* rsqrtf(double) :299 */
    double acc = 0.0;
    for (int64_t i = 0; i < n; i++) {
        /* `int qt = _round2int(v / s, rounding) + o;` with float o: int + float -> float -> int (:259) */
        int32_t qt = sat_f2i((float)oracle_round2int(x[i] / s, rounding) + o);
        float p;
        if (qt > qmax) { p = ((float)qmax - o) * dy[i]; grad_x[i] = 0; }
        else if (qt < qmin) { p = ((float)qmin - o) * dy[i]; grad_x[i] = 0; }
        else {
            float q = (float)(qt - sat_f2i(o)) * s; /* DequantizeScalar<int,float,int>: o converts to int */
            p = (q - x[i]) * dy[i] / s;
            grad_x[i] = dy[i];
        }
        acc += (double)p;
    }
    grad_s[0] = (float)(acc * (double)grad_factor);
}

/* QuantizeTensor_LC_B: linear.cu:326-433; grad_factor = rsqrt(n * clip_max) (:402) */
void oracle_fq_linear_c_bwd(const float* x, const float* dy, int64_t n, const float* scale,
                            const float* offset, int64_t num_channel, int64_t elem_per_channel,
                            int qmin, int qmax, int rounding, float* grad_x, float* grad_s) {
    float grad_factor = (float)(1.0 / sqrt((double)n * (double)qmax));
    double* acc = (double*)calloc((size_t)num_channel, sizeof(double));
    for (int64_t i = 0; i < n; i++) {
        int64_t c = (i / elem_per_channel) % num_channel;
        float s = scale[c];
        float o = roundf(offset[c]);
        int32_t qt = sat_f2i((float)oracle_round2int(x[i] / s, rounding) + o);
        float p;
        if (qt > qmax) { p = ((float)qmax - o) * dy[i]; grad_x[i] = 0; }
        else if (qt < qmin) { p = ((float)qmin - o) * dy[i]; grad_x[i] = 0; }
        else {
            float q = (float)(qt - sat_f2i(o)) * s;
            p = (q - x[i]) / s * dy[i];
            grad_x[i] = dy[i];
        }
        acc[c] += (double)p;
    }
    for (int64_t c = 0; c < num_channel; c++) grad_s[c] = (float)(acc[c] * (double)grad_factor);
    free(acc);
}

/* QuantizeScalarFloating: common.cuh:154-226 */
typedef union { float value; uint32_t data; } fp_bits;

float oracle_fq_float_quant_scalar(float value, float scale, int exponent, int mantissa,
                                   float clip_min, float clip_max, int rounding) {
    fp_bits helper, rounding_helper;
    float unscaled = value / scale;

    int32_t exponent_min = -(1 << (exponent - 1)) + 1;
    int32_t exponent_max = (1 << (exponent - 1));

    uint32_t fp32_sign = 0;
    int32_t fp32_exp = (exponent_max + 127) << 23;
    int32_t fp32_mantissa = ~(0x007FFFFF >> mantissa) & 0x007FFFFF;
    helper.data = fp32_sign + (uint32_t)fp32_mantissa + (uint32_t)fp32_exp;
    float theoretical_maximum = helper.value;

    float hi = clip_max < theoretical_maximum ? clip_max : theoretical_maximum;   /* min() :182 */
    float lo = clip_min > -theoretical_maximum ? clip_min : -theoretical_maximum; /* max() :184 */
    if (unscaled > hi) return hi;
    if (unscaled < lo) return lo;

    helper.value = unscaled;
    fp32_sign = helper.data & 0x80000000u;
    fp32_exp = (int32_t)(helper.data & 0x7F800000u);
    fp32_mantissa = (int32_t)(helper.data & 0x007FFFFFu);

    if (((fp32_exp >> 23) - 127) < exponent_min + 1) {
        float min_subnormal = 1.0f / (float)(1 << ((1 << (exponent - 1)) + mantissa - 2));
        return (float)oracle_round2int(unscaled / min_subnormal, rounding) * min_subnormal;
    }

    rounding_helper.data = (((uint32_t)fp32_mantissa << mantissa) & 0x007FFFFFu) + 0x3F800000u;
    uint32_t round_bit = (uint32_t)oracle_round2int(rounding_helper.value - 1, rounding);

    uint32_t m = ((((uint32_t)fp32_mantissa) >> (23 - mantissa)) + round_bit) << (23 - mantissa);
    helper.data = fp32_sign + m + (uint32_t)fp32_exp;

    float v = helper.value; /* CLIP<float> common.cuh:70-76 */
    if (v > clip_max) return clip_max;
    if (v < clip_min) return clip_min;
    return v;
}

/* QuantizeTensor_FT: floating.cu:36-75; offset only enters the dequant (q - o) * s */
void oracle_fq_float_t(const float* x, int64_t n, const float* scale, const float* offset,
                       int exponent, int mantissa, float clip_min, float clip_max,
                       int rounding, float* out) {
    float s = scale[0], o = offset[0];
    for (int64_t i = 0; i < n; i++) {
        float qt = oracle_fq_float_quant_scalar(x[i], s, exponent, mantissa, clip_min, clip_max, rounding);
        out[i] = (qt - o) * s;
    }
}

/* QuantizeTensor_FC: floating.cu:77-131 */
void oracle_fq_float_c(const float* x, int64_t n, const float* scale, const float* offset,
                       int64_t num_channel, int64_t elem_per_channel,
                       int exponent, int mantissa, float clip_min, float clip_max,
                       int rounding, float* out) {
    for (int64_t i = 0; i < n; i++) {
        int64_t c = (i / elem_per_channel) % num_channel;
        float qt = oracle_fq_float_quant_scalar(x[i], scale[c], exponent, mantissa, clip_min, clip_max, rounding);
        out[i] = (qt - offset[c]) * scale[c];
    }
}

/* QuantizeTensor_FT_B / _FC_B: floating.cu:133-331.  The reference divides by
 * sqrtf((float)(n * clip_max)) per block-partial; restated with a double accumulator. */
void oracle_fq_float_c_bwd(const float* x, const float* dy, int64_t n, const float* scale,
                           const float* offset, int64_t num_channel, int64_t elem_per_channel,
                           int exponent, int mantissa, float clip_min, float clip_max,
                           int rounding, float* grad_x, float* grad_s) {
    double* acc = (double*)calloc((size_t)num_channel, sizeof(double));
    float denom = sqrtf((float)((float)n * clip_max));
    for (int64_t i = 0; i < n; i++) {
        int64_t c = (i / elem_per_channel) % num_channel;
        float s = scale[c], inv_s = 1 / s, o = offset[c];
        float cmin = s * (clip_min - o), cmax = s * (clip_max - o);
        float qt = oracle_fq_float_quant_scalar(x[i], s, exponent, mantissa, clip_min - 1, clip_max + 1, rounding);
        float q = (qt - o) * s;
        float p;
        if (qt == clip_max + 1) { p = cmax * dy[i] * inv_s; grad_x[i] = 0; }
        else if (qt == clip_min - 1) { p = cmin * dy[i] * inv_s; grad_x[i] = 0; }
        else { p = (q - x[i]) * inv_s * dy[i]; grad_x[i] = dy[i]; }
        acc[c] += (double)p;
    }
    for (int64_t c = 0; c < num_channel; c++) grad_s[c] = (float)(acc[c] / (double)denom);
    free(acc);
}

/* _Histogram_T: sort.cu:75-89 (accumulates INTO hist) */
void oracle_hist_sym_t(const float* x, int64_t n, float hist_scale, int clip_outliers,
                       int32_t* hist, int64_t bins) {
    for (int64_t i = 0; i < n; i++) {
        int32_t b = sat_f2i(floorf(fabsf(x[i]) / hist_scale));
        if (clip_outliers && (int64_t)b > bins - 1) continue;
        else if ((int64_t)b > bins - 1) b = (int32_t)(bins - 1);
        hist[b] += 1;
    }
}

/* _Histogram_Asymmetric_T: sort.cu:113-139 */
void oracle_hist_asym_t(const float* x, int64_t n, float vmin, float vmax, int clip_outliers,
                        int32_t* hist, int64_t bins) {
    float hist_scale = (vmax - vmin) / (float)bins;
    for (int64_t i = 0; i < n; i++) {
        int32_t b = sat_f2i(floorf((x[i] - vmin) / hist_scale));
        if ((int64_t)b > bins - 1) { if (clip_outliers) continue; else b = (int32_t)(bins - 1); }
        if (b < 0) { if (clip_outliers) continue; else b = 0; }
        hist[b] += 1;
    }
}

/* _Histogram_C: sort.cu:167-185; hist is [C, bins] */
void oracle_hist_sym_c(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                       float hist_scale, int clip_outliers, int32_t* hist, int64_t bins) {
    for (int64_t i = 0; i < n; i++) {
        int32_t b = sat_f2i(floorf(fabsf(x[i]) / hist_scale));
        if (clip_outliers && (int64_t)b > bins - 1) continue;
        else if ((int64_t)b > bins - 1) b = (int32_t)(bins - 1);
        int64_t c = (i / elem_per_channel) % num_channel;
        hist[c * bins + b] += 1;
    }
}

static int cmp_float(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* Quantile_T: sort.cu:6-20, 42-59.  dest = [sorted[rn(n*q)], sorted[rn(n*(1-q))]] */
void oracle_quantile_t(const float* x, int64_t n, float q, float* dest) {
    float* s = (float*)malloc((size_t)n * sizeof(float));
    memcpy(s, x, (size_t)n * sizeof(float));
    qsort(s, (size_t)n, sizeof(float), cmp_float);
    int64_t max_pos = (int64_t)sat_f2i(nearbyintf((float)n * q));        /* __float2int_rn(n * q) */
    if (max_pos > n - 1) max_pos = n - 1; if (max_pos < 0) max_pos = 0;
    int64_t min_pos = (int64_t)sat_f2i(nearbyintf((float)n * (1 - q)));
    if (min_pos > n - 1) min_pos = n - 1; if (min_pos < 0) min_pos = 0;
    dest[0] = s[max_pos];
    dest[1] = s[min_pos];
    free(s);
}

/* Isotone_T: sort.cu:23-40, 61-73.  dest = [max, 2nd max, min, 2nd min] of the sorted copy */
void oracle_isotone_t(const float* x, int64_t n, float* dest) {
    float* s = (float*)malloc((size_t)n * sizeof(float));
    memcpy(s, x, (size_t)n * sizeof(float));
    qsort(s, (size_t)n, sizeof(float), cmp_float);
    if (n == 1) { dest[0] = dest[1] = dest[2] = dest[3] = s[0]; }
    else { dest[0] = s[n - 1]; dest[1] = s[n - 2]; dest[2] = s[0]; dest[3] = s[1]; }
    free(s);
}

/* per-tensor / per-channel running min & max: what TorchMinMaxObserver.observe collects,
 * ppq/quantization/observer/range.py:86-98 (torch.min / torch.max; NaN-free inputs) */
void oracle_minmax_t(const float* x, int64_t n, float* minmax) {
    float mn = minmax[0], mx = minmax[1];
    for (int64_t i = 0; i < n; i++) { if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; }
    minmax[0] = mn; minmax[1] = mx;
}
void oracle_minmax_c(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                     float* mins, float* maxs) {
    for (int64_t i = 0; i < n; i++) {
        int64_t c = (i / elem_per_channel) % num_channel;
        if (x[i] < mins[c]) mins[c] = x[i];
        if (x[i] > maxs[c]) maxs[c] = x[i];
    }
}

/* compute_mse_loss: ppq/csrc/cpu/hist_mse.cc:3-28 (float accumulation, the path taken
 * when USING_CUDA_KERNEL is True, observer/range.py:423-425) */
float oracle_mse_loss(const int64_t* hist, int64_t bins, int start, int step, int end) {
    int64_t num_of_elements = 0; float loss = 0.0f;
    for (int64_t i = 0; i < bins; i++) num_of_elements += hist[i];
    for (int idx = 0; idx < (int)bins; idx++) {
        float error = 0.0f;
        int64_t bin = hist[idx];
        if (idx < start) error = (float)(start - idx - 1 + 0.5);
        else if (idx > end) error = (float)(idx - end + 0.5);
        else {
            int64_t l_idx = (idx - start) % step;
            int64_t r_idx = step - l_idx - 1;
            if (l_idx == r_idx) error = (float)(l_idx + 0.25);
            else {
                float l_err = (float)(l_idx + 0.5);
                float r_err = (float)(r_idx + 0.5);
                error = l_err < r_err ? l_err : r_err;
            }
        }
        loss += ((float)bin * error * error) / (float)num_of_elements;
    }
    return loss;
}

/* the same loss in double: the pure-Python loop of observer/range.py:431-454 */
double oracle_mse_loss_f64(const int64_t* hist, int64_t bins, int start, int step, int end) {
    int64_t num_of_elements = 0; double loss = 0.0;
    for (int64_t i = 0; i < bins; i++) num_of_elements += hist[i];
    for (int idx = 0; idx < (int)bins; idx++) {
        double error;
        if (idx < start) error = (start - idx - 1) + 0.5;
        else if (idx > end) error = (idx - end) + 0.5;
        else {
            int l_idx = (idx - start) % step;
            int r_idx = step - l_idx - 1;
            if (l_idx == r_idx) error = l_idx + 0.25;
            else { double l = l_idx + 0.5, r = r_idx + 0.5; error = l < r ? l : r; }
        }
        loss += ((double)hist[idx] * error * error) / (double)num_of_elements;
    }
    return loss;
}

/* _TensorClip_T / _C: train.cu:35-113 */
void oracle_tensor_clip_t(const float* v, const float* ref, const float* limit, int64_t n, float* out) {
    float l = limit[0];
    for (int64_t i = 0; i < n; i++) {
        float lo = ref[i] - l, hi = ref[i] + l, x = v[i];
        out[i] = x > hi ? hi : (x < lo ? lo : x);
    }
}
void oracle_tensor_clip_c(const float* v, const float* ref, const float* limit, int64_t n,
                          int64_t num_channel, int64_t elem_per_channel, float* out) {
    for (int64_t i = 0; i < n; i++) {
        int64_t c = (i / elem_per_channel) % num_channel;
        float lo = ref[i] - limit[c], hi = ref[i] + limit[c], x = v[i];
        out[i] = x > hi ? hi : (x < lo ? lo : x);
    }
}

/* _RoundingLoss_LT / _LC: train.cu:115-175, 216-275 (double accumulator; compare with tolerance).
 * NB the kernels round the offset with nearbyint here, not round(). */
void oracle_rounding_loss_l(const float* x, int64_t n, const float* scale, const float* offset,
                            int64_t num_channel, int64_t elem_per_channel,
                            int qmin, int qmax, int rounding, float* out) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; i++) {
        int64_t c = num_channel > 0 ? (i / elem_per_channel) % num_channel : 0;
        float s = scale[c];
        int32_t o = sat_f2i(nearbyintf(offset[c]));
        float v = x[i];
        float dq = fq_linear_scalar(v, s, o, qmin, qmax, rounding);
        float diff = fabsf(dq - v);
        /* LT compares against s * (clip - o) with the INT offset; LC with the raw float offset */
        float ofs = num_channel > 0 ? offset[c] : (float)o;
        if (v > s * ((float)qmax - ofs)) diff = 0;
        if (v < s * ((float)qmin - ofs)) diff = 0;
        acc += (double)diff;
    }
    out[0] = (float)(acc / (double)sqrtf((float)n));
}

/* _RoundingLoss_LT_B / _LC_B: train.cu:177-214, 277-338 */
void oracle_rounding_loss_l_bwd(const float* x, const float* dy, int64_t n, const float* scale,
                                const float* offset, int64_t num_channel, int64_t elem_per_channel,
                                int qmin, int qmax, int rounding, float* dx) {
    float root = sqrtf((float)n);
    for (int64_t i = 0; i < n; i++) {
        int64_t c = num_channel > 0 ? (i / elem_per_channel) % num_channel : 0;
        float s = scale[c];
        int32_t o = sat_f2i(nearbyintf(offset[c]));
        float v = x[i];
        float dq = fq_linear_scalar(v, s, o, qmin, qmax, rounding);
        float grad = (float)((v > dq) ? 1 : -1) * dy[0];
        float ofs = num_channel > 0 ? offset[c] : (float)o;
        if (v > s * ((float)qmax - ofs)) grad = 0;
        if (v < s * ((float)qmin - ofs)) grad = 0;
        dx[i] = grad / root;
    }
}
