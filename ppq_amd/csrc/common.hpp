// common.hpp -- shared device helpers + host launch plumbing for libppq_hip.so (gfx950 only).
//
// Arithmetic contract (bit-exactness against the reference kernels, ppq/csrc/cuda/common.cuh):
//   * every translation unit is compiled with -ffp-contract=off: no FMA contraction anywhere;
//   * `a / b` on float is the correctly rounded IEEE quotient (hipcc's default
//     -fhip-fp32-correctly-rounded-divide-sqrt: v_div_scale / v_div_fmas / v_div_fixup);
//     the reference insists on true division too ("never do (1 / s)", linear.cu:73-74);
//   * float -> int32 is v_cvt_i32_f32: saturating, NaN -> 0 -- the same contract as the
//     cvt.rzi.s32.f32 the reference kernels compile to;
//   * round(x/s) + offset is a SATURATING int32 add (v_add_i32 clamp).  The reference adds with
//     wrap-around (UB in C++); both agree whenever the reference's add does not overflow, and
//     the saturating form agrees with the reference's PyTorch path in the overflow case.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ppq_hip.h"

namespace ppqhip {

constexpr int kWave = 64;            // wavefront width on gfx950
constexpr int kNumCU = 256;          // MI355X (SPX): sizes the persistent per-workgroup accumulators (an upper bound on grids)
int num_cu();                        // compute units of the CURRENT device (hipDeviceProp_t::multiProcessorCount, cached): a
                                     // partitioned device (CPX / DPX) gets proportionally smaller grids; never more than kNumCU
constexpr int kBlock = 256;          // default workgroup: 4 waves, one per SIMD

enum Rounding : int {
    ROUND_HALF_EVEN = 0, ROUND_HALF_UP = 1, ROUND_HALF_DOWN = 2, ROUND_HALF_TOWARDS_ZERO = 3,
    ROUND_HALF_FAR_FORM_ZERO = 4, ROUND_TO_NEAR_INT = 5, ROUND_UP = 6, ROUND_DOWN = 7
};

// ---- host side ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

// kernel ids for the profiling aid (ppqhip_prof_*)
enum KernelId : int {
    K_FQ_LINEAR_T = 0, K_FQ_LINEAR_C, K_FQ_LINEAR_T_BWD, K_FQ_LINEAR_C_BWD, K_FQ_FLOAT_T, K_FQ_FLOAT_C,
    K_FQ_FLOAT_BWD, K_HIST_SYM_T, K_HIST_ASYM_T, K_HIST_SYM_C, K_QUANTILE, K_ISOTONE, K_MINMAX_T,
    K_MINMAX_C, K_MSE_SEARCH, K_KL_LOSSES, K_TENSOR_CLIP, K_ROUNDING_LOSS, K_CHANNEL_SUM, K_FLOAT_SCALE_SEARCH, K_LSQ_FINISH, K_NUM
};
extern const char* const kKernelNames[K_NUM];

// RAII bracket around one logical kernel launch (possibly several device kernels): when
// profiling is on it records a hipEvent pair on `stream` and books `bytes` algorithmic bytes.
struct LaunchScope {
    LaunchScope(KernelId id, double bytes, hipStream_t stream);
    ~LaunchScope();
    int slot;
    hipStream_t stream;
};

int finish_launch(const char* what);   // hipGetLastError -> status
void* scratch(hipStream_t stream, size_t bytes);   // device scratch private to (device, stream)
constexpr size_t kZeroedArenaBytes = 32 * 16384 * sizeof(int);      // 32 partial rows of the largest LDS histogram (2 MB)
void* zeroed_arena(hipStream_t stream, size_t bytes);   // private to (device, stream), all zero between launches (users re-zero what they dirty); may return nullptr

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// division of a 31-bit numerator by an invariant divisor d >= 1 (Granlund-Montgomery):
// q = (mulhi(n, m) + n) >> l  with l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1.
struct FastDiv {
    uint32_t d, m, l;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f; f.d = d; f.l = 0;
    while ((1ull << f.l) < d) f.l++;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << f.l) - d)) / d + 1);
    return f;
}

// ---- device side ---------------------------------------------------------------------------
#if defined(__HIPCC__)

__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) {
    return (__umulhi(n, f.m) + n) >> f.l;
}

// Balanced split of T work items over G workgroups: workgroup g owns [begin, end); the first T % G workgroups own one item more.
// (floor(g * T / G) needs a 64-bit product and a 64-bit division: ~200 scalar instructions at the head of every persistent
// kernel, on the critical path of a latency-bound launch.  This is one 32-bit division.)
__device__ __forceinline__ void even_split(uint32_t T, uint32_t G, uint32_t g, uint32_t& begin, uint32_t& end) {
    const uint32_t q = T / G, rem = T - q * G;
    begin = g * q + min(g, rem);
    end = begin + q + (g < rem ? 1u : 0u);
}

// v_cvt_i32_f32: round-toward-zero, saturating, NaN -> 0 (the value passed in is already integral)
__device__ __forceinline__ int f2i_sat(float v) {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// v_cvt_i32_f64: same contract for double
__device__ __forceinline__ int d2i_sat(double v) {
    int r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

__device__ __forceinline__ int add_sat(int a, int b) { return __builtin_elementwise_add_sat(a, b); }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v > hi ? hi : (v < lo ? lo : v); }

// _round2int, common.cuh:88-114.  HALF_UP / HALF_DOWN evaluate `value + .5` in double exactly as
// the reference does (the literal .5 is a double there).
template <int R>
__device__ __forceinline__ int round2int_t(float value) {
    if constexpr (R == ROUND_HALF_EVEN) return f2i_sat(__builtin_rintf(value));
    else if constexpr (R == ROUND_HALF_UP) return d2i_sat(__builtin_floor((double)value + .5));
    else if constexpr (R == ROUND_HALF_DOWN) return d2i_sat(__builtin_ceil((double)value - .5));
    else if constexpr (R == ROUND_HALF_TOWARDS_ZERO)
        return value > 0 ? round2int_t<ROUND_HALF_DOWN>(value) : round2int_t<ROUND_HALF_UP>(value);
    else if constexpr (R == ROUND_HALF_FAR_FORM_ZERO)
        return value > 0 ? round2int_t<ROUND_HALF_UP>(value) : round2int_t<ROUND_HALF_DOWN>(value);
    else if constexpr (R == ROUND_UP) return f2i_sat(__builtin_ceilf(value));
    else if constexpr (R == ROUND_DOWN) return f2i_sat(__builtin_floorf(value));
    else return f2i_sat(__builtin_roundf(value));
}

__device__ __forceinline__ int round2int(float value, int rounding) {
    switch (rounding) {
        case ROUND_HALF_EVEN: return round2int_t<ROUND_HALF_EVEN>(value);
        case ROUND_HALF_UP: return round2int_t<ROUND_HALF_UP>(value);
        case ROUND_HALF_DOWN: return round2int_t<ROUND_HALF_DOWN>(value);
        case ROUND_HALF_TOWARDS_ZERO: return round2int_t<ROUND_HALF_TOWARDS_ZERO>(value);
        case ROUND_HALF_FAR_FORM_ZERO: return round2int_t<ROUND_HALF_FAR_FORM_ZERO>(value);
        case ROUND_UP: return round2int_t<ROUND_UP>(value);
        case ROUND_DOWN: return round2int_t<ROUND_DOWN>(value);
        default: return round2int_t<ROUND_TO_NEAR_INT>(value);
    }
}

// `int o = std::round(offset)`: linear.cu:52,76,149,175
__device__ __forceinline__ int round_offset(float o) { return f2i_sat(__builtin_roundf(o)); }

// QuantizeScalar + DequantizeScalar, common.cuh:116-147 / linear.cu:79-83
template <int R>
__device__ __forceinline__ float fq_linear_scalar(float x, float s, int o, int qmin, int qmax, int rounding) {
    const float qt = x / s;
    const int r = (R >= 0) ? round2int_t<(R >= 0 ? R : 0)>(qt) : round2int(qt, rounding);
    const int q = clampi(add_sat(r, o), qmin, qmax);
    return (float)(q - o) * s;
}

// ---- round-to-nearest-even quotient without the IEEE division on the common path ------------------------------------
// The reference divides (`value / scale`, linear.cu:73-74 "never do (1 / s)") and rounds half to even.  rint(RN(x / s)) is
// reproduced from t = RN(x * rc), rc ~ 1 / s, whenever that is PROVABLY the same integer:
//   * rc carries a relative error <= 2^-23 (v_rcp_f32: 1 ulp; the exactly rounded 1.0f / s: 2^-24), the product one rounding
//     (2^-24), the reference's quotient q one rounding (2^-24): |t - q| <= |t| * 2^-22 * (1 + 2^-22);
//   * rint(t) != rint(q) needs a half-integer h = k + 0.5 with min(t, q) <= h <= max(t, q) (a tie counts: q == h rounds to
//     even, t may sit on either side), i.e. within |t| * 2^-22 of t.  d = t - rint(t) is exact (|t| < 2^23, Sterbenz), so is
//     0.5 - |d| (a multiple of ulp(t), at most 0.5): the distance from t to the nearest half-integer.  Lanes with
//     |d| + |t| * 2^-20 <= 0.5 (four times the bound) are safe; the others -- about |t| * 2^-19 of all values, every
//     |t| >= 2^19 -- take the true division (see rne_tie_margin for NaN / inf).
//   * rc must be a normal number for the error bound to hold: scales outside [2^-100, 2^100] (and <= 0, NaN) hand back
//     NaN, which sends every lane through the division.
#ifndef PPQHIP_FQ_RCP
#define PPQHIP_FQ_RCP 1
#endif
__device__ __forceinline__ float fq_safe_rcp(float s) {
    const float a = __builtin_fabsf(s);
    return (a >= 0x1p-100f && a <= 0x1p100f) ? __builtin_amdgcn_rcpf(s) : __builtin_nanf("");
}
// distance-to-tie test as ONE number per lane: m = |t - rint(t)| + |t| 2^-20 (one fma; its single rounding is absorbed by the
// 4x margin: for |t| < 0.25 no tie is in reach at all, above that the rounding is < 3e-8 against a margin of |t| 7e-7).  The lane
// is safe when m <= 0.5; NaN compares unordered -> unsafe.  (v_max_f32 drops a NaN operand: a float4 with ONE NaN / inf element
// may therefore pass the combined test below -- for such an element both paths give the same result anyway: rint keeps NaN / inf,
// the saturating conversion maps them to 0 / INT_MAX, whether the quotient came from the reciprocal or from the division.)
__device__ __forceinline__ float rne_tie_margin(float t, float d) {             // d = t - rint(t)
    return __builtin_fmaf(__builtin_fabsf(t), 0x1p-20f, __builtin_fabsf(d));
}

// _round2int(x / s) for the four elements of a float4 (one scale): the reciprocal path above for ROUND_HALF_EVEN, the reference's
// division otherwise.  ONE divergent region per float4.
template <int R>
__device__ __forceinline__ void round_quotient4(const float4& a, float s, float rc, int rounding, int (&r)[4]) {
    if constexpr (R == ROUND_HALF_EVEN && PPQHIP_FQ_RCP != 0) {
        typedef float pk2 __attribute__((ext_vector_type(2)));                   // v_pk_mul_f32 / v_pk_add_f32: two lanes' worth per instruction
        const pk2 t01 = pk2{a.x, a.y} * rc, t23 = pk2{a.z, a.w} * rc;
        float r0 = __builtin_rintf(t01.x), r1 = __builtin_rintf(t01.y), r2 = __builtin_rintf(t23.x), r3 = __builtin_rintf(t23.y);
        const pk2 d01 = t01 - pk2{r0, r1}, d23 = t23 - pk2{r2, r3};
        const float m0 = rne_tie_margin(t01.x, d01.x), m1 = rne_tie_margin(t01.y, d01.y);
        const float m2 = rne_tie_margin(t23.x, d23.x), m3 = rne_tie_margin(t23.y, d23.y);
        if (!(__builtin_fmaxf(__builtin_fmaxf(m0, m1), __builtin_fmaxf(m2, m3)) <= 0.5f)) {
            // rare: some element sits next to a rounding tie (or the scale is degenerate: rc = NaN): the reference's own arithmetic
            if (!(m0 <= 0.5f)) r0 = __builtin_rintf(a.x / s);
            if (!(m1 <= 0.5f)) r1 = __builtin_rintf(a.y / s);
            if (!(m2 <= 0.5f)) r2 = __builtin_rintf(a.z / s);
            if (!(m3 <= 0.5f)) r3 = __builtin_rintf(a.w / s);
        }
        r[0] = f2i_sat(r0); r[1] = f2i_sat(r1); r[2] = f2i_sat(r2); r[3] = f2i_sat(r3);
    } else {
        r[0] = (R >= 0) ? round2int_t<(R >= 0 ? R : 0)>(a.x / s) : round2int(a.x / s, rounding);
        r[1] = (R >= 0) ? round2int_t<(R >= 0 ? R : 0)>(a.y / s) : round2int(a.y / s, rounding);
        r[2] = (R >= 0) ? round2int_t<(R >= 0 ? R : 0)>(a.z / s) : round2int(a.z / s, rounding);
        r[3] = (R >= 0) ? round2int_t<(R >= 0 ? R : 0)>(a.w / s) : round2int(a.w / s, rounding);
    }
}

// four elements of one (scale, offset): QuantizeScalar + DequantizeScalar
template <int R>
__device__ __forceinline__ float4 fq_linear4(const float4& a, float s, float rc, int o, int qmin, int qmax, int rounding) {
    int r[4];
    round_quotient4<R>(a, s, rc, rounding, r);
    float4 out;
    out.x = (float)(clampi(add_sat(r[0], o), qmin, qmax) - o) * s;
    out.y = (float)(clampi(add_sat(r[1], o), qmin, qmax) - o) * s;
    out.z = (float)(clampi(add_sat(r[2], o), qmin, qmax) - o) * s;
    out.w = (float)(clampi(add_sat(r[3], o), qmin, qmax) - o) * s;
    return out;
}

// wave64 reductions through DPP-backed shuffles
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// float atomic min / max on plain float storage (no NaNs): sign-split integer atomics.  The split is
// on the SIGN BIT, not on `v >= 0`: -0.0f compares >= 0 but its pattern 0x80000000 is INT_MIN, and a
// signed atomicMin with it would overwrite any stored negative minimum.
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
    if (__float_as_int(v) >= 0) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    if (__float_as_int(v) >= 0) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

typedef float v4f __attribute__((ext_vector_type(4)));

// 16-B load; NT = streaming ("nontemporal") hint: the line is not kept in L2 / Infinity Cache.
template <bool NT>
__device__ __forceinline__ float4 load4(const float4* p) {
    if (NT) {
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    }
    return *p;
}

// Stream every element of x[0..n) through f(value).  Each workgroup owns ONE CONTIGUOUS chunk of
// the tensor (better DRAM-page / TLB locality than a grid-strided interleave: measured 5.5 vs
// 4.7 TB/s on a 205 MB read), walks it in tiles of blockDim * U float4 with U 16-B loads in flight
// per lane; the tail (n % 4) and unaligned tensors fall back to 4-B loads.  Order is unspecified.
template <int U, bool NT = false, typename F>
__device__ __forceinline__ void stream_elems(const float* __restrict__ x, uint32_t n, bool vec_ok, F f,
                                             uint32_t bidx = blockIdx.x, uint32_t nblk = gridDim.x) {
    // bidx / nblk: index of this workgroup among the nblk workgroups sharing the tensor
    uint32_t done = 0;
    if (vec_ok) {
        const uint32_t nvec = n >> 2;
        const float4* xv = reinterpret_cast<const float4*>(x);
        const uint32_t tile = blockDim.x * U;
        const uint32_t tiles = (nvec + tile - 1) / tile;
        const uint32_t per = (tiles + nblk - 1) / nblk;                      // tiles per workgroup
        const uint32_t lo = bidx * per * tile;
        const uint32_t hi = min(lo + per * tile, nvec);
        for (uint32_t v = lo + threadIdx.x; v < hi; v += tile) {
            float4 a[U];
#pragma unroll
            for (int k = 0; k < U; k++)
                if (v + k * blockDim.x < hi) a[k] = load4<NT>(&xv[v + k * blockDim.x]);
#pragma unroll
            for (int k = 0; k < U; k++)
                if (v + k * blockDim.x < hi) { f(a[k].x); f(a[k].y); f(a[k].z); f(a[k].w); }
        }
        done = nvec << 2;
    }
    const uint32_t stride = nblk * blockDim.x;
    for (uint32_t i = done + bidx * blockDim.x + threadIdx.x; i < n; i += stride) f(x[i]);
}

// Same traversal with a WAVE-UNIFORM trip count (ballot-safe): on_tile(sample, valid) is called by
// every lane once per tile before its elements, on_elem(value, valid) for every slot of the tile
// (valid == false for slots past the end); the scalar remainder reports one element per "tile".
template <int U, typename FT, typename FE>
__device__ __forceinline__ void stream_tiles(const float* __restrict__ x, uint32_t n, bool vec_ok, FT on_tile,
                                             FE on_elem, uint32_t bidx = blockIdx.x, uint32_t nblk = gridDim.x) {
    uint32_t done = 0;
    if (vec_ok) {
        const uint32_t nvec = n >> 2;
        const float4* xv = reinterpret_cast<const float4*>(x);
        const uint32_t tile = blockDim.x * U;
        const uint32_t tiles = (nvec + tile - 1) / tile;
        const uint32_t per = (tiles + nblk - 1) / nblk;
        const uint32_t hi = min((bidx + 1) * per * tile, nvec);
        uint32_t v = bidx * per * tile + threadIdx.x;
        for (uint32_t t = 0; t < per; t++, v += tile) {
            float4 a[U];
#pragma unroll
            for (int k = 0; k < U; k++)
                a[k] = (v + k * blockDim.x < hi) ? xv[v + k * blockDim.x] : make_float4(0.f, 0.f, 0.f, 0.f);
            on_tile(a[0].x, v < hi);
#pragma unroll
            for (int k = 0; k < U; k++) {
                const bool in = v + k * blockDim.x < hi;
                on_elem(a[k].x, in); on_elem(a[k].y, in); on_elem(a[k].z, in); on_elem(a[k].w, in);
            }
        }
        done = nvec << 2;
    }
    const uint32_t stride = nblk * blockDim.x;
    const uint32_t rem = n - done;
    const uint32_t trips = (rem + stride - 1) / stride;
    uint32_t i = bidx * blockDim.x + threadIdx.x;
    for (uint32_t t = 0; t < trips; t++, i += stride) {
        const bool in = i < rem;
        const float a = in ? x[done + i] : 0.f;
        if ((t & 15u) == 0) on_tile(a, in);
        on_elem(a, in);
    }
}

#ifndef PPQHIP_HIST_ASM
#define PPQHIP_HIST_ASM 1              // EXEC-mask commits in inline assembly (WaveBinCounter::commit4_exec); 0 = compiler-generated
#endif
constexpr int kHotMinLanes = 12;       // lanes that must share the candidate bin to make it hot

// One wavefront's counting state over an LDS histogram (`h` = the copy this wave adds into, `last` = bins - 1):
// predicated ds_add_u32 commits plus a wave-uniform "hot bin" whose hits are counted on the scalar unit (a bin
// shared by many lanes would serialise the LDS atomic unit: a k-way same-address ds_add costs ~k cycles).
// ASYM: bins may be negative; CLIP: out-of-range bins are dropped (else clamped); HOT: hot-bin register on.
// Used by the value histograms (hist.hip: Binner) and the radix pass of the quantile (reduce.hip).
template <bool ASYM, bool CLIP, bool HOT>
struct WaveBinCounter {
    int* h;
    int last;        // bins - 1
    int hot_bin;     // wave-uniform, -1 = none
    int hot_cnt;     // wave-uniform: hits are counted with ballot + s_bcnt1 (no VALU)

    __device__ __forceinline__ void init(int* copy, int bins) { h = copy; last = bins - 1; hot_bin = -1; hot_cnt = 0; }

    // population count of a lane mask as a 32-bit SCALAR (two s_bcnt1_i32_b32: a 64-bit count would be
    // compared / selected on the vector unit, which drags hot_bin / hot_cnt into VGPRs)
    static __device__ __forceinline__ int popc_mask(unsigned long long m) {
        return __builtin_popcount((unsigned)m) + __builtin_popcount((unsigned)(m >> 32));
    }

    // count one value in bin b (as produced by bins4 / bin1); `in` = the value exists.
    template <bool IN_ALWAYS>
    __device__ __forceinline__ void commit(int b, bool in) {
        bool ok = IN_ALWAYS ? true : in;
        if (CLIP) ok = ok && (unsigned)b <= (unsigned)last;            // b < 0 wraps above `last`
        else b = ASYM ? (b < 0 ? 0 : (b > last ? last : b)) : (b > last ? last : b);
        if (HOT) {
            const bool hit = b == hot_bin;
            // wave-uniform count on the scalar unit: s_and + s_bcnt1 + s_add, no VALU
            hot_cnt += popc_mask(__builtin_amdgcn_ballot_w64(ok) & __builtin_amdgcn_ballot_w64(hit));
            if (ok && !hit) atomicAdd(&h[b], 1);
        } else {
            if (ok) atomicAdd(&h[b], 1);
        }
    }

    // Four full-wave commits (CLIP && HOT, every lane holds a value) with the EXEC mask doing the
    // predication: per element v_cmpx (b <= last narrows EXEC), v_cmp (hot hits -> VCC, counted with
    // s_bcnt1 on the scalar unit and removed from EXEC), v_lshl_add (LDS address), ds_add_u32 --
    // 3 VALU + 4 SALU instead of the ~6 + ~8 the compiler needs for the same logic through lane masks.
    // Only SGPR / VCC / EXEC hand-offs that the hardware interlocks are used (VALU-written VCC is read by
    // SALU only; EXEC is restored from an SGPR pair before control returns to compiled code).
    __device__ __forceinline__ void commit4_exec(const int (&b)[4]) {
#if PPQHIP_HIST_ASM
        unsigned long long save;
        int t0, a0r;
        int cnt = __builtin_amdgcn_readfirstlane(hot_cnt);
        const int hot = __builtin_amdgcn_readfirstlane(hot_bin), lastu = __builtin_amdgcn_readfirstlane(last);
        const unsigned base = (unsigned)(uintptr_t)h;
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "v_cmpx_ge_u32_e32 vcc, %[last], %[b0]\n\t"
            "v_cmp_eq_u32_e32 vcc, %[hot], %[b0]\n\t"
            "s_bcnt1_i32_b64 %[t], vcc\n\t"
            "s_andn2_b64 exec, exec, vcc\n\t"
            "s_add_i32 %[cnt], %[cnt], %[t]\n\t"
            "v_lshl_add_u32 %[a], %[b0], 2, %[base]\n\t"
            "ds_add_u32 %[a], %[one]\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "v_cmpx_ge_u32_e32 vcc, %[last], %[b1]\n\t"
            "v_cmp_eq_u32_e32 vcc, %[hot], %[b1]\n\t"
            "s_bcnt1_i32_b64 %[t], vcc\n\t"
            "s_andn2_b64 exec, exec, vcc\n\t"
            "s_add_i32 %[cnt], %[cnt], %[t]\n\t"
            "v_lshl_add_u32 %[a], %[b1], 2, %[base]\n\t"
            "ds_add_u32 %[a], %[one]\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "v_cmpx_ge_u32_e32 vcc, %[last], %[b2]\n\t"
            "v_cmp_eq_u32_e32 vcc, %[hot], %[b2]\n\t"
            "s_bcnt1_i32_b64 %[t], vcc\n\t"
            "s_andn2_b64 exec, exec, vcc\n\t"
            "s_add_i32 %[cnt], %[cnt], %[t]\n\t"
            "v_lshl_add_u32 %[a], %[b2], 2, %[base]\n\t"
            "ds_add_u32 %[a], %[one]\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "v_cmpx_ge_u32_e32 vcc, %[last], %[b3]\n\t"
            "v_cmp_eq_u32_e32 vcc, %[hot], %[b3]\n\t"
            "s_bcnt1_i32_b64 %[t], vcc\n\t"
            "s_andn2_b64 exec, exec, vcc\n\t"
            "s_add_i32 %[cnt], %[cnt], %[t]\n\t"
            "v_lshl_add_u32 %[a], %[b3], 2, %[base]\n\t"
            "ds_add_u32 %[a], %[one]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [cnt] "+s"(cnt), [sv] "=&s"(save), [t] "=&s"(t0), [a] "=&v"(a0r)
            : [last] "s"(lastu), [hot] "s"(hot), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]),
              [base] "v"(base), [one] "v"(1)
            : "vcc", "scc", "memory");
        hot_cnt = cnt;
#else
        commit<true>(b[0], true); commit<true>(b[1], true); commit<true>(b[2], true); commit<true>(b[3], true);
#endif
    }

    __device__ __forceinline__ void flush_hot() {
        if (!HOT) return;
        if ((threadIdx.x & 63) == 0 && hot_cnt != 0 && hot_bin >= 0) atomicAdd(&h[hot_bin], hot_cnt);
        hot_cnt = 0;
    }

    // Re-elect the hot bin from one bin per lane; all lanes of the wave call this together.  The bin of
    // the first valid lane becomes hot when at least kHotMinLanes lanes share it.
    __device__ __forceinline__ void elect(int b, bool in) {
        if (!HOT) return;
        const bool valid = in && (unsigned)b <= (unsigned)last;
        const unsigned long long act = __builtin_amdgcn_ballot_w64(valid);
        if (act == 0ull) return;
        const int cand = __builtin_amdgcn_readlane(b, __builtin_ctzll(act));
        if (cand == hot_bin) return;
        const int share = popc_mask(act & __builtin_amdgcn_ballot_w64(b == cand));
        if (share >= kHotMinLanes) { flush_hot(); hot_bin = cand; }
    }
};

// One wavefront's view of an LDS counter array (`nbins` counters followed by 64 per-lane trash
// slots) with a "hot bin" kept in registers: a key shared by many lanes would serialise the LDS
// atomic unit (a k-way same-address ds_add costs ~k cycles), so hits on the elected hot bin are
// counted in a VGPR and flushed with one wave reduction.  add() is unconditional: pass trash() for
// "do not count".  elect() must be reached by all lanes of the wave together.
struct HotCounter {
    unsigned int* h;
    int nbins, hot_bin, hot_cnt;
    __device__ __forceinline__ void init(unsigned int* base, int bins) { h = base; nbins = bins; hot_bin = -1; hot_cnt = 0; }
    __device__ __forceinline__ int trash() const { return nbins + (int)(threadIdx.x & 63); }
    __device__ __forceinline__ void flush() {
        int c = hot_cnt;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) c += __shfl_xor(c, m, 64);
        if ((threadIdx.x & 63) == 0 && c != 0 && hot_bin >= 0) atomicAdd(&h[hot_bin], (unsigned int)c);
        hot_cnt = 0;
    }
    __device__ __forceinline__ void elect(int b, bool valid, int min_lanes = 12) {
        valid = valid && b < nbins;
        const unsigned long long act = __ballot(valid);
        if (act == 0ull) return;
        const int cand = __builtin_amdgcn_readlane(b, __ffsll((long long)act) - 1);
        if (cand == hot_bin) return;
        if ((int)__popcll(__ballot(valid && b == cand)) >= min_lanes) { flush(); hot_bin = cand; }
    }
    __device__ __forceinline__ void add(int slot) {
        const bool hit = slot == hot_bin;
        hot_cnt += hit ? 1 : 0;
        atomicAdd(&h[hit ? trash() : slot], 1u);
    }
};

#endif  // __HIPCC__

// grid size for a streaming kernel that consumes `work_items` items, `per_block` per block-pass,
// capped so that the chip holds every block at once (8 x 256-thread blocks per CU).
inline int stream_grid(int64_t work_items, int64_t per_block, int max_blocks = 0) {
    if (max_blocks <= 0) max_blocks = num_cu() * 8;
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

}  // namespace ppqhip
