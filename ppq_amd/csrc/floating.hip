// floating.hip -- low-precision float (FP8 E4M3 / E5M2 / generic E,M) fake-quant for gfx950.
//
// Replaces ppq/csrc/cuda/floating.cu; the scalar algorithm restates QuantizeScalarFloating
// (ppq/csrc/cuda/common.cuh:154-226) bit for bit, including its quirk that an exact mantissa tie
// under ROUND_HALF_EVEN rounds toward zero (nearbyint(0.5) == 0), which is NOT IEEE / OCP RNE --
// so the hardware v_cvt_pk_fp8_f32 conversions are deliberately not used here.
// Same streaming structure as linear.hip (one contiguous tile of float4s per workgroup).
#include "common.hpp"

#include <vector>

namespace ppqhip {

struct FloatFmt {
    float hi, lo;          // min(clip_max, theoretical_max), max(clip_min, -theoretical_max)
    float clip_min, clip_max;
    int exponent_min_p1;   // exponent_min + 1
    float min_subnormal;
    int mantissa;
    // fast path (quant_float_rne_pow2)
    uint32_t half_m1;      // (1 << (22 - mantissa)) - 1: added to |u|'s pattern, a carry out of the dropped bits <=> they exceed one half
    uint32_t keep;         // ~((1 << (23 - mantissa)) - 1)
    uint32_t sub_limit;    // patterns of |u| below this take the subnormal branch: (exponent_min + 1 + 127) << 23
    float sub_scale;       // 1 / min_subnormal (a power of two)
};

static int make_fmt(int exponent, int mantissa, float clip_min, float clip_max, FloatFmt* f, const char* what) {
    if (exponent <= 0 || exponent > 8 || mantissa < 0 || mantissa > 23) {
        set_error("%s: unsupported float format E%dM%d", what, exponent, mantissa);
        return PPQHIP_ERR_INVALID_VALUE;
    }
    const int sub_shift = (1 << (exponent - 1)) + mantissa - 2;
    if (sub_shift < 0 || sub_shift > 30) {
        // `1 << sub_shift` overflows int in the reference (common.cuh:210): undefined there.
        set_error("%s: E%dM%d needs 1 << %d, which the reference cannot represent", what, exponent, mantissa,
                  sub_shift);
        return PPQHIP_ERR_UNSUPPORTED;
    }
    const int exponent_min = -(1 << (exponent - 1)) + 1;
    const int exponent_max = (1 << (exponent - 1));
    union { float v; uint32_t d; } h;
    h.d = (uint32_t)((exponent_max + 127) << 23) + (uint32_t)(~(0x007FFFFF >> mantissa) & 0x007FFFFF);
    const float theo = h.v;
    f->hi = clip_max < theo ? clip_max : theo;
    f->lo = clip_min > -theo ? clip_min : -theo;
    f->clip_min = clip_min; f->clip_max = clip_max;
    f->exponent_min_p1 = exponent_min + 1;
    f->min_subnormal = 1.0f / (float)(1 << sub_shift);
    f->mantissa = mantissa;
    f->half_m1 = mantissa <= 22 ? (1u << (22 - mantissa)) - 1u : 0u;
    f->keep = mantissa <= 22 ? ~((1u << (23 - mantissa)) - 1u) : 0xFFFFFFFFu;
    f->sub_limit = (uint32_t)(exponent_min + 1 + 127) << 23;
    f->sub_scale = (float)(1 << sub_shift);
    return PPQHIP_OK;
}

// QuantizeScalarFloating, common.cuh:154-226 (the format-dependent constants are hoisted into fmt)
template <int R>
__device__ __forceinline__ float quant_float_scalar(float value, float scale, const FloatFmt& fmt, int rounding) {
    const float u = value / scale;
    if (u > fmt.hi) return fmt.hi;
    if (u < fmt.lo) return fmt.lo;
    const uint32_t bits = __float_as_uint(u);
    const uint32_t sign = bits & 0x80000000u;
    const int32_t exp = (int32_t)(bits & 0x7F800000u);
    const uint32_t man = bits & 0x007FFFFFu;
    if (((exp >> 23) - 127) < fmt.exponent_min_p1) {
        const float t = u / fmt.min_subnormal;
        const int r = (R >= 0) ? round2int_t<(R >= 0 ? R : 0)>(t) : round2int(t, rounding);
        return (float)r * fmt.min_subnormal;
    }
    const float frac = __uint_as_float(((man << fmt.mantissa) & 0x007FFFFFu) + 0x3F800000u) - 1;
    const uint32_t round_bit = (uint32_t)((R >= 0) ? round2int_t<(R >= 0 ? R : 0)>(frac) : round2int(frac, rounding));
    const uint32_t m = ((man >> (23 - fmt.mantissa)) + round_bit) << (23 - fmt.mantissa);
    const float v = __uint_as_float(sign + m + (uint32_t)exp);
    return v > fmt.clip_max ? fmt.clip_max : (v < fmt.clip_min ? fmt.clip_min : v);
}

// The same function for the case every shipped FP8 configuration is in -- ROUND_HALF_EVEN and a power-of-two scale
// (FP8Quantizer.py:99,196; the `floating` observer picks from {2^-7 .. 64}, observer/floating.py:97) -- without the two
// IEEE divisions and the data-dependent branches, bit for bit (tests: all 2^32 input patterns):
//   * value / 2^e == value * 2^-e (one exact scaling, rounded once either way; NaN payloads, infinities, underflow alike);
//     u / min_subnormal likewise (min_subnormal = 2^-k);
//   * round2int(frac) under HALF_EVEN with frac in [0, 1) is 1 iff the dropped mantissa bits exceed one half (a tie goes
//     to 0 = the reference's round-toward-zero-on-ties quirk), i.e. a carry out of `dropped + half - 1`; adding on the
//     whole magnitude pattern lets the carry run into the exponent exactly as `sign + m + exp` does (also for the
//     all-ones NaN pattern, where both wrap through the sign bit: hence `+ sign`, not `| sign`);
//   * (float)(int)rint(t) == rint(t) + 0.0f for |t| < 2^31 (the int round trip only loses the sign of a zero).
__device__ __forceinline__ uint32_t pow2_reciprocal_bits(float s) {        // 0 when s is not 2^e with 2^-e normal too
    const uint32_t b = __float_as_uint(s), e = b >> 23;
    return ((b & 0x807FFFFFu) == 0u && e >= 1u && e <= 253u) ? ((254u - e) << 23) : 0u;
}
__device__ __forceinline__ float quant_float_rne_pow2(float value, float rcp, const FloatFmt& fmt) {
    const float u = value * rcp;
    const uint32_t bits = __float_as_uint(u), sign = bits & 0x80000000u, mag = bits & 0x7FFFFFFFu;
    float vn = __uint_as_float(((mag + fmt.half_m1) & fmt.keep) + sign);
    vn = vn > fmt.clip_max ? fmt.clip_max : (vn < fmt.clip_min ? fmt.clip_min : vn);
    const float vs = (__builtin_rintf(u * fmt.sub_scale) + 0.0f) * fmt.min_subnormal;
    float v = mag < fmt.sub_limit ? vs : vn;
    v = u < fmt.lo ? fmt.lo : v;
    v = u > fmt.hi ? fmt.hi : v;
    return v;
}
template <int R>
__device__ __forceinline__ bool float_fast_ok(const FloatFmt& fmt) { return R == ROUND_HALF_EVEN && fmt.mantissa <= 22; }

// one contiguous tile of kBlock * U float4 per workgroup (see linear.hip)
template <int R, int U, bool NT>
__global__ __launch_bounds__(kBlock) void fq_float_t_tile_kernel(
    const float4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    float4* __restrict__ out, uint32_t nvec, const float* __restrict__ xtail, float* __restrict__ otail,
    int ntail, FloatFmt fmt, int rounding) {
    const uint32_t base = blockIdx.x * (kBlock * U) + threadIdx.x;
    float4 a[U];
#pragma unroll
    for (int k = 0; k < U; k++)
        a[k] = load4<NT>(&x[min(base + k * kBlock, nvec - 1)]);   // branch-free (clamped) so all U loads issue back to back
    const float s = scale[0], o = offset[0];
    const uint32_t rb = float_fast_ok<R>(fmt) ? pow2_reciprocal_bits(s) : 0u;
    if (rb) {                                                     // kernel-uniform
        const float rcp = __uint_as_float(rb);
#pragma unroll
        for (int k = 0; k < U; k++) {                             // unconditional arithmetic, predicated store (see linear.hip)
            float4 r;
            r.x = (quant_float_rne_pow2(a[k].x, rcp, fmt) - o) * s;
            r.y = (quant_float_rne_pow2(a[k].y, rcp, fmt) - o) * s;
            r.z = (quant_float_rne_pow2(a[k].z, rcp, fmt) - o) * s;
            r.w = (quant_float_rne_pow2(a[k].w, rcp, fmt) - o) * s;
            if (base + k * kBlock < nvec) out[base + k * kBlock] = r;
        }
        if (blockIdx.x == 0 && (int)threadIdx.x < ntail)
            otail[threadIdx.x] = (quant_float_rne_pow2(xtail[threadIdx.x], rcp, fmt) - o) * s;
        return;
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
        if (base + k * kBlock < nvec) {
            float4 r;
            r.x = (quant_float_scalar<R>(a[k].x, s, fmt, rounding) - o) * s;
            r.y = (quant_float_scalar<R>(a[k].y, s, fmt, rounding) - o) * s;
            r.z = (quant_float_scalar<R>(a[k].z, s, fmt, rounding) - o) * s;
            r.w = (quant_float_scalar<R>(a[k].w, s, fmt, rounding) - o) * s;
            out[base + k * kBlock] = r;
        }
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail)
        otail[threadIdx.x] = (quant_float_scalar<R>(xtail[threadIdx.x], s, fmt, rounding) - o) * s;
}

// generic (scalar) per-tensor / per-channel kernel; num_channel.d == 0 selects per tensor
template <int R>
__global__ __launch_bounds__(kBlock) void fq_float_scalar_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    float* __restrict__ out, uint32_t n, FastDiv elem_per_channel, FastDiv num_channel, int per_channel,
    FloatFmt fmt, int rounding) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t c = 0;
        if (per_channel) {
            const uint32_t row = fdiv(i, elem_per_channel);
            c = row - fdiv(row, num_channel) * num_channel.d;
        }
        const float s = scale[c], o = offset[c];
        out[i] = (quant_float_scalar<R>(x[i], s, fmt, rounding) - o) * s;
    }
}

template <int R, int U, bool NT>
__global__ __launch_bounds__(kBlock) void fq_float_c_tile_kernel(
    const float4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    float4* __restrict__ out, uint32_t nvec, FastDiv vec_per_channel, FastDiv num_channel, FloatFmt fmt,
    int rounding) {
    const uint32_t base = blockIdx.x * (kBlock * U) + threadIdx.x;
    float4 a[U];
    float s[U], o[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        const uint32_t vv = min(base + k * kBlock, nvec - 1);      // clamped: branch-free loads
        {
            a[k] = load4<NT>(&x[vv]);
            const uint32_t row = fdiv(vv, vec_per_channel);
            const uint32_t c = row - fdiv(row, num_channel) * num_channel.d;
            s[k] = scale[c];
            o[k] = offset[c];
        }
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
        const uint32_t vv = base + k * kBlock;
        const uint32_t rb = float_fast_ok<R>(fmt) ? pow2_reciprocal_bits(s[k]) : 0u;
        if (rb) {                                                  // diverges only where a wave straddles channels of both kinds
            const float rcp = __uint_as_float(rb);                 // unconditional arithmetic, predicated store (see linear.hip)
            float4 r;
            r.x = (quant_float_rne_pow2(a[k].x, rcp, fmt) - o[k]) * s[k];
            r.y = (quant_float_rne_pow2(a[k].y, rcp, fmt) - o[k]) * s[k];
            r.z = (quant_float_rne_pow2(a[k].z, rcp, fmt) - o[k]) * s[k];
            r.w = (quant_float_rne_pow2(a[k].w, rcp, fmt) - o[k]) * s[k];
            if (vv < nvec) out[vv] = r;
        } else if (vv < nvec) {
            float4 r;
            r.x = (quant_float_scalar<R>(a[k].x, s[k], fmt, rounding) - o[k]) * s[k];
            r.y = (quant_float_scalar<R>(a[k].y, s[k], fmt, rounding) - o[k]) * s[k];
            r.z = (quant_float_scalar<R>(a[k].z, s[k], fmt, rounding) - o[k]) * s[k];
            r.w = (quant_float_scalar<R>(a[k].w, s[k], fmt, rounding) - o[k]) * s[k];
            out[vv] = r;
        }
    }
}

// QuantizeTensor_FT_B / _FC_B, floating.cu:133-331: STE + scale gradient with +-1 sentinel clip.
// The reference adds every block-partial divided by sqrtf((float)(n * clip_max)) with one atomic per block; here the
// tensor is cut into (row, chunk) workgroups of one channel each -- rows of elem_per_channel contiguous elements, chunks of
// 4096 -- that leave ONE partial sum per workgroup, and a second launch adds a channel's partials in a fixed order and
// divides once (deterministic; the first version issued one device atomic PER ELEMENT: 44 ms for [32, 512, 56, 56]).
__device__ __forceinline__ float fq_float_bwd_elem(float v, float d, float s, float inv_s, float o, float cmin, float cmax,
                                                   const FloatFmt& fmt_wide, float clip_min, float clip_max, int rounding,
                                                   float* gx) {
    const float qt = quant_float_scalar<-1>(v, s, fmt_wide, rounding);
    const float q = (qt - o) * s;
    if (qt == clip_max + 1) { *gx = 0.f; return cmax * d * inv_s; }
    if (qt == clip_min - 1) { *gx = 0.f; return cmin * d * inv_s; }
    *gx = d;
    return (q - v) * inv_s * d;
}

constexpr uint32_t kFloatBwdChunk = 4096;      // elements per workgroup
template <bool NT>
__global__ __launch_bounds__(kBlock) void fq_float_bwd_row_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    const float* __restrict__ dy, float* __restrict__ gx, float* __restrict__ partial, uint32_t epc, int vec_ok,
    FastDiv chunks, FastDiv num_channel, FloatFmt fmt_wide, float clip_min, float clip_max, int rounding) {
    __shared__ float lds[kBlock / kWave];
    const uint32_t row = fdiv(blockIdx.x, chunks);
    const uint32_t chunk = blockIdx.x - row * chunks.d;
    const uint32_t c = row - fdiv(row, num_channel) * num_channel.d;
    const float s = scale[c], inv_s = 1 / s, o = offset[c];
    const float cmin = s * (clip_min - o), cmax = s * (clip_max - o);
    const uint32_t lo = chunk * kFloatBwdChunk, hi = min(lo + kFloatBwdChunk, epc);
    const size_t base = (size_t)row * epc;
    float acc = 0.f;
    if (vec_ok) {                                  // epc % 4 == 0, 16-B aligned bases: 4 (x, dy) load pairs in flight per lane
        const float4* xv = reinterpret_cast<const float4*>(x + base);
        const float4* dv = reinterpret_cast<const float4*>(dy + base);
        float4* gv = reinterpret_cast<float4*>(gx + base);
        const uint32_t v1 = hi >> 2;
        for (uint32_t v = (lo >> 2) + threadIdx.x; v < v1; v += kBlock * 4) {
            float4 a[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t at = min(v + u * kBlock, v1 - 1); a[u] = load4<NT>(&xv[at]); d[u] = load4<NT>(&dv[at]); }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (v + u * kBlock >= v1) break;
                float4 g;
                acc += fq_float_bwd_elem(a[u].x, d[u].x, s, inv_s, o, cmin, cmax, fmt_wide, clip_min, clip_max, rounding, &g.x);
                acc += fq_float_bwd_elem(a[u].y, d[u].y, s, inv_s, o, cmin, cmax, fmt_wide, clip_min, clip_max, rounding, &g.y);
                acc += fq_float_bwd_elem(a[u].z, d[u].z, s, inv_s, o, cmin, cmax, fmt_wide, clip_min, clip_max, rounding, &g.z);
                acc += fq_float_bwd_elem(a[u].w, d[u].w, s, inv_s, o, cmin, cmax, fmt_wide, clip_min, clip_max, rounding, &g.w);
                gv[v + u * kBlock] = g;
            }
        }
    } else {
        for (uint32_t j = lo + threadIdx.x; j < hi; j += kBlock) {
            float g;
            acc += fq_float_bwd_elem(x[base + j], dy[base + j], s, inv_s, o, cmin, cmax, fmt_wide, clip_min, clip_max, rounding, &g);
            gx[base + j] = g;
        }
    }
    acc = wave_sum(acc);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) lds[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < kBlock / kWave; w++) t += lds[w];
        partial[blockIdx.x] = t;
    }
}

// short rows (elem_per_channel < 64: [N, C] matrices, channel-last views): grid-stride over elements, per-channel sums in
// LDS (num_channel <= 8192) or straight global atomics, one flush per workgroup
__global__ __launch_bounds__(kBlock) void fq_float_bwd_generic_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    const float* __restrict__ dy, float* __restrict__ gx, float* __restrict__ gs, uint32_t n, FastDiv elem_per_channel,
    FastDiv num_channel, int use_lds, FloatFmt fmt_wide, float clip_min, float clip_max, float denom, int rounding) {
    extern __shared__ float acc_lds[];
    const uint32_t C = num_channel.d;
    if (use_lds) {
        for (uint32_t c = threadIdx.x; c < C; c += kBlock) acc_lds[c] = 0.f;
        __syncthreads();
    }
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t row = fdiv(i, elem_per_channel);
        const uint32_t c = row - fdiv(row, num_channel) * C;
        const float s = scale[c], inv_s = 1 / s, o = offset[c];
        float g;
        const float p = fq_float_bwd_elem(x[i], dy[i], s, inv_s, o, s * (clip_min - o), s * (clip_max - o), fmt_wide, clip_min,
                                          clip_max, rounding, &g);
        gx[i] = g;
        if (use_lds) atomicAdd(&acc_lds[c], p);
        else atomicAdd(&gs[c], p / denom);
    }
    if (use_lds) {
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < C; c += kBlock) {
            const float v = acc_lds[c];
            if (v != 0.f) atomicAdd(&gs[c], v / denom);
        }
    }
}

// grad_s[c] = (sum of the partials of channel c: rows c, c + C, .. x their chunks, in index order) / denom
__global__ __launch_bounds__(kBlock) void fq_float_bwd_finish_kernel(const float* __restrict__ partial, uint32_t rows, uint32_t chunks,
                                                                     uint32_t C, float denom, float* __restrict__ gs) {
    __shared__ double lds[kBlock / kWave];
    const uint32_t c = blockIdx.x;
    const uint32_t per_channel = (rows / C) * chunks;                 // partials of this channel
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < per_channel; i += 8 * kBlock) {          // 8 loads in flight per lane
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t at = i + u * kBlock, r = at / chunks, k = at - r * chunks;
            v[u] = at < per_channel ? partial[(size_t)(r * C + c) * chunks + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += (double)v[u];
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kBlock / kWave; w++) t += lds[w];
        gs[c] = (float)t / denom;
    }
}

// ---- many tensors, one launch (the design of fq_linear_multi_kernel, linear.hip) ---------------------------
// The TRT_FP8 policy fake-quantises the weight of every Conv / Gemm / MatMul per channel on every forward
// (ViT-B/16: 50 weights): one launch serves them all.  Device job table + workgroup-count prefix in the arguments.
constexpr int kFqFloatMultiMax = 128;
struct FqFloatJob {
    const float* x;
    float* out;
    const float* scale;
    const float* offset;
    uint32_t n;
    uint32_t vec_ok;
    FastDiv per;          // vec_ok: float4 per channel row; else elements per channel row
    FastDiv nc;
    FloatFmt fmt;
};
struct FqFloatMultiArgs {
    uint32_t first_block[kFqFloatMultiMax];
    uint32_t count;
    int rounding;
    const FqFloatJob* jobs;
};

template <int R, int U>
__global__ __launch_bounds__(kBlock) void fq_float_multi_kernel(const FqFloatMultiArgs args) {
    uint32_t lo = 0, hi = args.count;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (args.first_block[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    const FqFloatJob j = args.jobs[lo];                 // uniform address: scalar loads
    const uint32_t base = (blockIdx.x - args.first_block[lo]) * (kBlock * U) + threadIdx.x;
    const uint32_t C = j.nc.d;
    auto one = [&](float v, float s, float o, uint32_t rb) {
        const float q = rb ? quant_float_rne_pow2(v, __uint_as_float(rb), j.fmt) : quant_float_scalar<R>(v, s, j.fmt, args.rounding);
        return (q - o) * s;
    };
    if (j.vec_ok) {
        const uint32_t nvec = j.n >> 2;
        const float4* xv = reinterpret_cast<const float4*>(j.x);
        float4* ov = reinterpret_cast<float4*>(j.out);
        float4 a[U];
        float s[U], o[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const uint32_t vv = min(base + k * kBlock, nvec - 1);
            a[k] = xv[vv];
            const uint32_t row = fdiv(vv, j.per);
            const uint32_t c = row - fdiv(row, j.nc) * C;
            s[k] = j.scale[c];
            o[k] = j.offset[c];
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            const uint32_t vv = base + k * kBlock;
            if (vv < nvec) {
                const uint32_t rb = float_fast_ok<R>(j.fmt) ? pow2_reciprocal_bits(s[k]) : 0u;
                float4 r;
                r.x = one(a[k].x, s[k], o[k], rb); r.y = one(a[k].y, s[k], o[k], rb);
                r.z = one(a[k].z, s[k], o[k], rb); r.w = one(a[k].w, s[k], o[k], rb);
                ov[vv] = r;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < U; k++) {
            const uint32_t e0 = (base + k * kBlock) * 4;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t e = e0 + q;
                if (e < j.n) {
                    const uint32_t row = fdiv(e, j.per);
                    const uint32_t c = row - fdiv(row, j.nc) * C;
                    const float sc = j.scale[c];
                    j.out[e] = one(j.x[e], sc, j.offset[c], float_fast_ok<R>(j.fmt) ? pow2_reciprocal_bits(sc) : 0u);
                }
            }
        }
    }
}

// ---- the scale search of the FP8 'floating' observer, batched ------------------------------------------------
// DirectMSEObserver (observer/floating.py:88-143) fake-quantises its collected values with each of 7 candidate
// scales and keeps the one with the least mean squared error: 7 x (quantise, subtract, square, mean) launches and a
// host synchronisation PER CONFIG -- 148 configs on ViT-B/16, 29 ms of a 100 ms calibration.  Here ONE launch
// serves every config of the graph: a workgroup owns one row (a per-tensor collection, or one channel of a weight),
// reads it once, evaluates all candidates on the value in registers and writes the squared-error sums (double,
// fixed summation order: deterministic).  The dequantised value is formed in float32 exactly as the fake-quant
// kernels form it, so err = float32 result of fq - x as the reference's `qt - fp`.
constexpr int kSearchMaxCandidates = 8;
constexpr int kSearchMaxJobs = 128;
struct FloatSearchJob {
    const float* x;       // rows x row_len, contiguous
    uint32_t row_len;
    FloatFmt fmt;
};
struct FloatSearchArgs {
    uint32_t first_row[kSearchMaxJobs];      // prefix of row counts: workgroup -> job
    uint32_t count;
    int rounding, num_candidates;
    float candidate[kSearchMaxCandidates];
    const FloatSearchJob* jobs;
    double* out;                              // [total rows][num_candidates]
    uint32_t out_row0;
};

template <int R>
__global__ __launch_bounds__(kBlock) void float_scale_search_kernel(const FloatSearchArgs args) {
    __shared__ double red[kSearchMaxCandidates][kBlock / kWave];
    uint32_t lo = 0, hi = args.count;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (args.first_row[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    const FloatSearchJob j = args.jobs[lo];
    const float* __restrict__ row = j.x + (size_t)(blockIdx.x - args.first_row[lo]) * j.row_len;
    double acc[kSearchMaxCandidates];
    uint32_t rb[kSearchMaxCandidates];
#pragma unroll
    for (int c = 0; c < kSearchMaxCandidates; c++) {
        acc[c] = 0.0;
        rb[c] = (c < args.num_candidates && float_fast_ok<R>(j.fmt)) ? pow2_reciprocal_bits(args.candidate[c]) : 0u;
    }
    for (uint32_t i = threadIdx.x; i < j.row_len; i += kBlock) {
        const float v = row[i];
#pragma unroll
        for (int c = 0; c < kSearchMaxCandidates; c++) {
            if (c < args.num_candidates) {
                const float s = args.candidate[c];
                const float q = rb[c] ? quant_float_rne_pow2(v, __uint_as_float(rb[c]), j.fmt)
                                      : quant_float_scalar<R>(v, s, j.fmt, args.rounding);
                const float e = (q - 0.0f) * s - v;          // offset 0 (floating.py:112,129); float32 like the fake-quant output
                acc[c] += (double)e * (double)e;
            }
        }
    }
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < kSearchMaxCandidates; c++) {
        double v = acc[c];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) red[c][wid] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < args.num_candidates) {
        double v = 0.0;
        for (int w = 0; w < kBlock / kWave; w++) v += red[threadIdx.x][w];
        args.out[(size_t)(args.out_row0 + blockIdx.x) * args.num_candidates + threadIdx.x] = v;
    }
}

static int validate(int64_t n, const char* what) {
    if (n <= 0) { set_error("%s: tensor is empty", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (n > 0x7fffffffLL) { set_error("%s: too many elements", what); return PPQHIP_ERR_INVALID_VALUE; }
    return PPQHIP_OK;
}

constexpr int64_t kStreamElems = 48ll << 20;   // >= 192 MiB: streaming loads (see linear.hip)
constexpr int kTileU = 2;
constexpr int kSmallU = 1;                     // latency-bound tensors: twice the waves, half the work behind each load (linear.hip)
constexpr int64_t kSmallElems = 4ll << 20;

template <int R>
static void launch_ft(const float* x, const float* scale, const float* offset, float* out, int64_t n,
                      const FloatFmt& fmt, int rounding, hipStream_t st) {
    if (aligned16(x) && aligned16(out) && n >= 4) {
        const uint32_t nvec = (uint32_t)(n >> 2);
        const int ntail = (int)(n & 3);
        const float* xt = x + (size_t)nvec * 4;
        float* ot = out + (size_t)nvec * 4;
#define PPQ_LAUNCH_FT(U, NT)                                                                                          \
        hipLaunchKernelGGL((fq_float_t_tile_kernel<R, U, NT>), dim3((nvec + kBlock * U - 1) / (kBlock * U)), dim3(kBlock), 0, st, \
                           (const float4*)x, scale, offset, (float4*)out, nvec, xt, ot, ntail, fmt, rounding)
        if (n >= kStreamElems) PPQ_LAUNCH_FT(kTileU, true);
        else if (n <= kSmallElems) PPQ_LAUNCH_FT(kSmallU, false);
        else PPQ_LAUNCH_FT(kTileU, false);
#undef PPQ_LAUNCH_FT
    } else {
        hipLaunchKernelGGL((fq_float_scalar_kernel<R>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, st, x, scale,
                           offset, out, (uint32_t)n, make_fastdiv(1), make_fastdiv(1), 0, fmt, rounding);
    }
}

template <int R>
static void launch_fc(const float* x, const float* scale, const float* offset, float* out, int64_t n, int64_t C,
                      int64_t epc, const FloatFmt& fmt, int rounding, hipStream_t st) {
    if (aligned16(x) && aligned16(out) && epc % 4 == 0) {
        const uint32_t nvec = (uint32_t)(n >> 2);
        const FastDiv vpc = make_fastdiv((uint32_t)(epc / 4)), nc = make_fastdiv((uint32_t)C);
#define PPQ_LAUNCH_FC(U, NT)                                                                                          \
        hipLaunchKernelGGL((fq_float_c_tile_kernel<R, U, NT>), dim3((nvec + kBlock * U - 1) / (kBlock * U)), dim3(kBlock), 0, st, \
                           (const float4*)x, scale, offset, (float4*)out, nvec, vpc, nc, fmt, rounding)
        if (n >= kStreamElems) PPQ_LAUNCH_FC(kTileU, true);
        else if (n <= kSmallElems) PPQ_LAUNCH_FC(kSmallU, false);
        else PPQ_LAUNCH_FC(kTileU, false);
#undef PPQ_LAUNCH_FC
    } else {
        hipLaunchKernelGGL((fq_float_scalar_kernel<R>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, st, x, scale,
                           offset, out, (uint32_t)n, make_fastdiv((uint32_t)epc), make_fastdiv((uint32_t)C), 1, fmt,
                           rounding);
    }
}

}  // namespace ppqhip

using namespace ppqhip;

extern "C" {

int ppqhip_fq_float_t(const float* x, const float* scale, const float* offset, float* out, int64_t n,
                      int exponent, int mantissa, float clip_min, float clip_max, int rounding,
                      void* stream) {
    if (int st = validate(n, "fq_float_t")) return st;
    FloatFmt fmt;
    if (int st = make_fmt(exponent, mantissa, clip_min, clip_max, &fmt, "fq_float_t")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_FQ_FLOAT_T, 8.0 * (double)n, s);
    if (rounding == ROUND_HALF_EVEN) launch_ft<ROUND_HALF_EVEN>(x, scale, offset, out, n, fmt, rounding, s);
    else launch_ft<-1>(x, scale, offset, out, n, fmt, rounding, s);
    return finish_launch("fq_float_t");
}

int ppqhip_fq_float_c(const float* x, const float* scale, const float* offset, float* out, int64_t n,
                      int64_t num_channel, int64_t elem_per_channel, int exponent, int mantissa,
                      float clip_min, float clip_max, int rounding, void* stream) {
    if (int st = validate(n, "fq_float_c")) return st;
    if (num_channel <= 0 || elem_per_channel <= 0 || n % (num_channel * elem_per_channel) != 0) {
        set_error("fq_float_c: bad channel geometry"); return PPQHIP_ERR_INVALID_VALUE;
    }
    FloatFmt fmt;
    if (int st = make_fmt(exponent, mantissa, clip_min, clip_max, &fmt, "fq_float_c")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_FQ_FLOAT_C, 8.0 * (double)n, s);
    if (rounding == ROUND_HALF_EVEN)
        launch_fc<ROUND_HALF_EVEN>(x, scale, offset, out, n, num_channel, elem_per_channel, fmt, rounding, s);
    else launch_fc<-1>(x, scale, offset, out, n, num_channel, elem_per_channel, fmt, rounding, s);
    return finish_launch("fq_float_c");
}

int64_t ppqhip_fq_float_multi_table_bytes(int num_jobs) {
    return num_jobs > 0 ? (int64_t)sizeof(FqFloatJob) * num_jobs : 0;
}

int ppqhip_fq_float_multi(const ppqhip_fq_float_job* jobs, int num_jobs, int rounding, void* device_table, int upload,
                          void* stream) {
    if (num_jobs <= 0) return PPQHIP_OK;
    if (jobs == nullptr || device_table == nullptr) {
        set_error("fq_float_multi: jobs / device_table is null"); return PPQHIP_ERR_INVALID_VALUE;
    }
    hipStream_t s = (hipStream_t)stream;
    double bytes = 0.0;
    for (int k = 0; k < num_jobs; k++) {
        const ppqhip_fq_float_job& j = jobs[k];
        if (int st = validate(j.n, "fq_float_multi")) return st;
        if (j.num_channel <= 0 || j.elem_per_channel <= 0 || j.n % (j.num_channel * j.elem_per_channel) != 0) {
            set_error("fq_float_multi: job %d has a bad channel geometry", k); return PPQHIP_ERR_INVALID_VALUE;
        }
        if (!j.x || !j.out || !j.scale || !j.offset) {
            set_error("fq_float_multi: job %d has a null pointer", k); return PPQHIP_ERR_INVALID_VALUE;
        }
        FloatFmt probe;
        if (int st = make_fmt(j.exponent, j.mantissa, j.clip_min, j.clip_max, &probe, "fq_float_multi")) return st;
        bytes += 8.0 * (double)j.n;
    }
    LaunchScope scope(K_FQ_FLOAT_C, bytes, s);
    constexpr int U = 2;
    for (int base = 0; base < num_jobs; base += kFqFloatMultiMax) {
        const int count = (num_jobs - base) < kFqFloatMultiMax ? (num_jobs - base) : kFqFloatMultiMax;
        FqFloatMultiArgs args;
        args.count = (uint32_t)count; args.rounding = rounding;
        args.jobs = (const FqFloatJob*)device_table + base;
        std::vector<FqFloatJob> table(upload ? count : 0);
        uint32_t blocks = 0;
        for (int k = 0; k < count; k++) {
            const ppqhip_fq_float_job& src = jobs[base + k];
            const bool vec = aligned16(src.x) && aligned16(src.out) && (src.elem_per_channel % 4 == 0);
            args.first_block[k] = blocks;
            const uint64_t quads = ((uint64_t)src.n + 3) / 4;
            blocks += (uint32_t)((quads + kBlock * U - 1) / (kBlock * U));
            if (upload) {
                FqFloatJob& d = table[k];
                d.x = src.x; d.out = src.out; d.scale = src.scale; d.offset = src.offset;
                d.n = (uint32_t)src.n; d.vec_ok = vec ? 1u : 0u;
                d.per = make_fastdiv((uint32_t)(vec ? src.elem_per_channel / 4 : src.elem_per_channel));
                d.nc = make_fastdiv((uint32_t)src.num_channel);
                make_fmt(src.exponent, src.mantissa, src.clip_min, src.clip_max, &d.fmt, "fq_float_multi");
            }
        }
        if (upload) {
            // pageable source: the runtime stages the copy before returning, `table` may go out of scope
            if (int st = check_hip(hipMemcpyAsync((FqFloatJob*)device_table + base, table.data(), sizeof(FqFloatJob) * count,
                                                  hipMemcpyHostToDevice, s), "fq_float_multi table upload"))
                return st;
        }
        if (rounding == ROUND_HALF_EVEN)
            hipLaunchKernelGGL((fq_float_multi_kernel<ROUND_HALF_EVEN, U>), dim3(blocks), dim3(kBlock), 0, s, args);
        else
            hipLaunchKernelGGL((fq_float_multi_kernel<-1, U>), dim3(blocks), dim3(kBlock), 0, s, args);
    }
    return finish_launch("fq_float_multi");
}

int64_t ppqhip_float_scale_search_table_bytes(int num_jobs) {
    return num_jobs > 0 ? (int64_t)sizeof(FloatSearchJob) * num_jobs : 0;
}

int ppqhip_float_scale_search(const ppqhip_float_search_job* jobs, int num_jobs, const float* candidates, int num_candidates,
                              int rounding, void* device_table, double* out, void* stream) {
    if (num_jobs <= 0) return PPQHIP_OK;
    if (jobs == nullptr || candidates == nullptr || device_table == nullptr || out == nullptr) {
        set_error("float_scale_search: null argument"); return PPQHIP_ERR_INVALID_VALUE;
    }
    if (num_candidates <= 0 || num_candidates > kSearchMaxCandidates) {
        set_error("float_scale_search: 1 .. %d candidate scales are supported, got %d", kSearchMaxCandidates, num_candidates);
        return PPQHIP_ERR_INVALID_VALUE;
    }
    hipStream_t s = (hipStream_t)stream;
    double bytes = 0.0;
    for (int k = 0; k < num_jobs; k++) {
        const ppqhip_float_search_job& j = jobs[k];
        if (j.x == nullptr || j.rows <= 0 || j.row_len <= 0 || j.rows > 0x7fffffffLL || j.row_len > 0x7fffffffLL) {
            set_error("float_scale_search: job %d is empty or too large", k); return PPQHIP_ERR_INVALID_VALUE;
        }
        FloatFmt probe;
        if (int st = make_fmt(j.exponent, j.mantissa, j.clip_min, j.clip_max, &probe, "float_scale_search")) return st;
        bytes += 4.0 * (double)j.rows * (double)j.row_len;
    }
    LaunchScope scope(K_FLOAT_SCALE_SEARCH, bytes, s);
    uint32_t row0 = 0;
    for (int base = 0; base < num_jobs; base += kSearchMaxJobs) {
        const int count = (num_jobs - base) < kSearchMaxJobs ? (num_jobs - base) : kSearchMaxJobs;
        FloatSearchArgs args;
        args.count = (uint32_t)count; args.rounding = rounding; args.num_candidates = num_candidates;
        for (int c = 0; c < kSearchMaxCandidates; c++) args.candidate[c] = c < num_candidates ? candidates[c] : 1.0f;
        args.jobs = (const FloatSearchJob*)device_table + base;
        args.out = out; args.out_row0 = row0;
        std::vector<FloatSearchJob> table(count);
        uint32_t rows = 0;
        for (int k = 0; k < count; k++) {
            const ppqhip_float_search_job& src = jobs[base + k];
            args.first_row[k] = rows;
            rows += (uint32_t)src.rows;
            table[k].x = src.x; table[k].row_len = (uint32_t)src.row_len;
            make_fmt(src.exponent, src.mantissa, src.clip_min, src.clip_max, &table[k].fmt, "float_scale_search");
        }
        if (int st = check_hip(hipMemcpyAsync((FloatSearchJob*)device_table + base, table.data(), sizeof(FloatSearchJob) * count,
                                              hipMemcpyHostToDevice, s), "float_scale_search table upload"))
            return st;
        if (rounding == ROUND_HALF_EVEN) hipLaunchKernelGGL((float_scale_search_kernel<ROUND_HALF_EVEN>), dim3(rows), dim3(kBlock), 0, s, args);
        else hipLaunchKernelGGL((float_scale_search_kernel<-1>), dim3(rows), dim3(kBlock), 0, s, args);
        row0 += rows;
    }
    return finish_launch("float_scale_search");
}

int ppqhip_fq_float_c_bwd(const float* x, const float* scale, const float* offset, const float* grad_y,
                          float* grad_x, float* grad_s, int64_t n, int64_t num_channel,
                          int64_t elem_per_channel, int exponent, int mantissa, float clip_min,
                          float clip_max, int rounding, void* stream) {
    if (int st = validate(n, "fq_float_c_bwd")) return st;
    if (num_channel <= 0 || elem_per_channel <= 0 || n % (num_channel * elem_per_channel) != 0) {
        set_error("fq_float_c_bwd: bad channel geometry"); return PPQHIP_ERR_INVALID_VALUE;
    }
    FloatFmt fmt;   // the backward quantises against [clip_min - 1, clip_max + 1]: floating.cu:163-164
    if (int st = make_fmt(exponent, mantissa, clip_min - 1, clip_max + 1, &fmt, "fq_float_c_bwd")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_FQ_FLOAT_BWD, 12.0 * (double)n, s);
    const float denom = sqrtf((float)((float)n * clip_max));
    if (elem_per_channel < 64) {
        if (int st = check_hip(hipMemsetAsync(grad_s, 0, sizeof(float) * (size_t)num_channel, s), "memset grad_s")) return st;
        const int use_lds = num_channel <= 8192;
        hipLaunchKernelGGL(fq_float_bwd_generic_kernel, dim3(stream_grid(n, kBlock * 8, num_cu() * 2)), dim3(kBlock),
                           use_lds ? sizeof(float) * (size_t)num_channel : 0, s, x, scale, offset, grad_y, grad_x, grad_s, (uint32_t)n,
                           make_fastdiv((uint32_t)elem_per_channel), make_fastdiv((uint32_t)num_channel), use_lds, fmt, clip_min,
                           clip_max, denom, rounding);
        return finish_launch("fq_float_c_bwd");
    }
    const uint32_t chunks = (uint32_t)((elem_per_channel + kFloatBwdChunk - 1) / kFloatBwdChunk);
    const int64_t rows = n / elem_per_channel;
    if (rows * chunks > 0x7fffffffLL) { set_error("fq_float_c_bwd: too many rows"); return PPQHIP_ERR_INVALID_VALUE; }
    float* partial = (float*)scratch(s, sizeof(float) * (size_t)(rows * chunks));
    if (partial == nullptr) return PPQHIP_ERR_HIP;
    const int vec_ok = (elem_per_channel % 4 == 0 && aligned16(x) && aligned16(grad_y) && aligned16(grad_x)) ? 1 : 0;
    // x and dy together exceed cache residency from 96 MiB each: streaming (nontemporal) loads, as the linear backward kernels
    if (n >= (24ll << 20))
        hipLaunchKernelGGL(fq_float_bwd_row_kernel<true>, dim3((uint32_t)(rows * chunks)), dim3(kBlock), 0, s, x, scale, offset, grad_y,
                           grad_x, partial, (uint32_t)elem_per_channel, vec_ok, make_fastdiv(chunks),
                           make_fastdiv((uint32_t)num_channel), fmt, clip_min, clip_max, rounding);
    else
        hipLaunchKernelGGL(fq_float_bwd_row_kernel<false>, dim3((uint32_t)(rows * chunks)), dim3(kBlock), 0, s, x, scale, offset, grad_y,
                           grad_x, partial, (uint32_t)elem_per_channel, vec_ok, make_fastdiv(chunks),
                           make_fastdiv((uint32_t)num_channel), fmt, clip_min, clip_max, rounding);
    hipLaunchKernelGGL(fq_float_bwd_finish_kernel, dim3((uint32_t)num_channel), dim3(kBlock), 0, s, (const float*)partial,
                       (uint32_t)rows, chunks, (uint32_t)num_channel, denom, grad_s);
    return finish_launch("fq_float_c_bwd");
}

}  // extern "C"
