// linear.hip -- integer fake-quant kernels (forward + LSQ backward) for gfx950.
//
// Replaces ppq/csrc/cuda/linear.cu.  Design (not a translation of the CUDA launch shapes):
//   * HBM-bound streaming: 16 B per lane per access (global_load_dwordx4 / global_store_dwordx4),
//     one wavefront = 1 KiB per instruction, 2 independent accesses in flight per lane;
//   * one contiguous tile of 512 float4 per 256-thread workgroup and as many workgroups as tiles:
//     on MI355X this streams faster than a chip-sized persistent grid-stride loop, and tiny
//     tensors still spread over every CU;
//   * per-tensor scale / offset live in SGPRs (uniform scalar loads);
//   * per-channel: the channel of a float4 is found with a multiply-high division by an
//     invariant (no integer divide in the loop); scale/offset gathers hit L1/L2;
//   * the tail (n % 4) and unaligned tensors run through the same arithmetic in a scalar kernel,
//     so results do not depend on the launch shape.
#include <cstdlib>

#include <vector>

#include "common.hpp"

namespace ppqhip {

// --------------------------------------------------------------------------- per tensor forward
// One tile per workgroup, no loop: workgroup b owns float4s [b * kBlock * U, (b + 1) * kBlock * U).
// Tens of thousands of short-lived workgroups over contiguous tiles stream faster than a
// persistent grid-stride loop on MI355X (6.8 vs 4.9 TB/s on 205 MB in + 205 MB out).  NT selects
// streaming loads for tensors that cannot be cache resident anyway; stores stay cacheable because
// the next operation of the graph consumes the output.
template <int R, int U, bool NT>
__global__ __launch_bounds__(kBlock) void fq_linear_t_tile_kernel(
    const float4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    float4* __restrict__ out, uint32_t nvec, const float* __restrict__ xtail, float* __restrict__ otail,
    int ntail, int qmin, int qmax, int rounding) {
    const uint32_t base = blockIdx.x * (kBlock * U) + threadIdx.x;
    float4 a[U];
#pragma unroll
    for (int k = 0; k < U; k++)
        a[k] = load4<NT>(&x[min(base + k * kBlock, nvec - 1)]);   // branch-free (clamped) so all U loads issue back to back
    const float s = scale[0];
    const int o = round_offset(offset[0]);
    const float rc = fq_safe_rcp(s);
    // the arithmetic is unconditional (lanes past the end recompute the clamped element), only the STORE is predicated: with
    // the whole body under `if (v < nvec)` the compiler sinks the loads into the branch, behind the wait for the scale
#pragma unroll
    for (int k = 0; k < U; k++) {
        const float4 r = fq_linear4<R>(a[k], s, rc, o, qmin, qmax, rounding);
        if (base + k * kBlock < nvec) out[base + k * kBlock] = r;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail)
        otail[threadIdx.x] = fq_linear_scalar<R>(xtail[threadIdx.x], s, o, qmin, qmax, rounding);
}

template <int R, int U, bool NT>
__global__ __launch_bounds__(kBlock) void fq_linear_c_tile_kernel(
    const float4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    float4* __restrict__ out, uint32_t nvec, FastDiv vec_per_channel, FastDiv num_channel,
    int qmin, int qmax, int rounding) {
    const uint32_t base = blockIdx.x * (kBlock * U) + threadIdx.x;
    float4 a[U];
    float s[U];
    int o[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        const uint32_t vv = min(base + k * kBlock, nvec - 1);      // clamped: branch-free loads
        {
            a[k] = load4<NT>(&x[vv]);
            const uint32_t row = fdiv(vv, vec_per_channel);
            const uint32_t c = row - fdiv(row, num_channel) * num_channel.d;
            s[k] = scale[c];
            o[k] = round_offset(offset[c]);
        }
    }
    // unconditional arithmetic, predicated store: under `if (vv < nvec)` the compiler sinks the tile's x / scale loads into the
    // branch, BEHIND the wait for the offsets (two dependent memory round trips on a latency-bound tensor)
#pragma unroll
    for (int k = 0; k < U; k++) {
        const uint32_t vv = base + k * kBlock;
        const float4 r = fq_linear4<R>(a[k], s[k], fq_safe_rcp(s[k]), o[k], qmin, qmax, rounding);
        if (vv < nvec) out[vv] = r;
    }
}

template <int R>
__global__ __launch_bounds__(kBlock) void fq_linear_t_scalar_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    float* __restrict__ out, uint32_t n, int qmin, int qmax, int rounding) {
    const float s = scale[0];
    const int o = round_offset(offset[0]);
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = fq_linear_scalar<R>(x[i], s, o, qmin, qmax, rounding);
}

// --------------------------------------------------------------------------- per channel forward
// (vector path: fq_linear_c_tile_kernel above; elem_per_channel % 4 == 0 so a float4 never straddles channels)
template <int R>
__global__ __launch_bounds__(kBlock) void fq_linear_c_scalar_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    float* __restrict__ out, uint32_t n, FastDiv elem_per_channel, FastDiv num_channel,
    int qmin, int qmax, int rounding) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t row = fdiv(i, elem_per_channel);
        const uint32_t c = row - fdiv(row, num_channel) * num_channel.d;
        out[i] = fq_linear_scalar<R>(x[i], scale[c], round_offset(offset[c]), qmin, qmax, rounding);
    }
}

// --------------------------------------------------------------------------- quantise to integer
// PPQLinearQuant_toInt, ppq/quantization/qfunction/linear.py:218-238 (the reference does this with torch ops; it
// has no kernel for it): q = clamp(ppq_tensor_round(x / s) + o, qmin, qmax) in FLOAT32 -- the RAW offset, not the
// rounded one the fake-quant kernels use -- then .type(int8 | uint8 | int32), i.e. truncation towards zero.
// ppq_tensor_round (utils/round.py:9-49) is a float32 formula per policy, not common.cuh's _round2int.
__device__ __forceinline__ float tensor_round_f32(float v, int rounding) {
    const float sgn = v > 0.f ? 1.f : (v < 0.f ? -1.f : v);           // torch.sign: 0 -> 0, NaN -> NaN
    switch (rounding) {
        case ROUND_HALF_EVEN: return __builtin_rintf(v);
        case ROUND_UP: return __builtin_ceilf(v);
        case ROUND_HALF_TOWARDS_ZERO: return sgn * __builtin_ceilf(__builtin_fabsf(v) - 0.5f);
        case ROUND_HALF_FAR_FORM_ZERO: return sgn * __builtin_floorf(__builtin_fabsf(v) + 0.5f);
        case ROUND_HALF_DOWN: return __builtin_ceilf(v - 0.5f);
        default: return __builtin_floorf(v + 0.5f);                    // ROUND_HALF_UP
    }
}
__device__ __forceinline__ int to_int_scalar(float x, float s, float o, float qmin, float qmax, int rounding) {
    float t = tensor_round_f32(x / s, rounding) + o;
    t = t < qmin ? qmin : t;            // torch.clamp: NaN stays NaN (both compares false)
    t = t > qmax ? qmax : t;
    return f2i_sat(t);                  // truncates; NaN -> 0
}
// OUT: int8_t / uint8_t / int32_t.  Workgroup b owns elements [b * kBlock * 4, ..): one 16-B load, one 4- or 16-B store per lane.
template <typename OUT, bool CHANNEL>
__global__ __launch_bounds__(kBlock) void to_int_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ offset, OUT* __restrict__ out, uint32_t n,
                                                        int vec_ok, FastDiv elem_per_channel, FastDiv num_channel, float qmin,
                                                        float qmax, int rounding) {
    const uint32_t i0 = (blockIdx.x * kBlock + threadIdx.x) * 4u;
    if (i0 >= n) return;
    auto so = [&](uint32_t i, float& s, float& o) {
        uint32_t c = 0;
        if (CHANNEL) { const uint32_t row = fdiv(i, elem_per_channel); c = row - fdiv(row, num_channel) * num_channel.d; }
        s = scale[c]; o = offset[c];
    };
    if (vec_ok && i0 + 4 <= n) {        // elem_per_channel % 4 == 0: the four share a channel
        const float4 a = *reinterpret_cast<const float4*>(x + i0);
        float s, o;
        so(i0, s, o);
        const int q0 = to_int_scalar(a.x, s, o, qmin, qmax, rounding), q1 = to_int_scalar(a.y, s, o, qmin, qmax, rounding);
        const int q2 = to_int_scalar(a.z, s, o, qmin, qmax, rounding), q3 = to_int_scalar(a.w, s, o, qmin, qmax, rounding);
        if (sizeof(OUT) == 1) {
            *reinterpret_cast<uint32_t*>(out + i0) = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
        } else {
            *reinterpret_cast<int4*>(out + i0) = make_int4(q0, q1, q2, q3);
        }
    } else {
        for (uint32_t i = i0; i < min(i0 + 4u, n); i++) {
            float s, o;
            so(i, s, o);
            out[i] = (OUT)to_int_scalar(x[i], s, o, qmin, qmax, rounding);
        }
    }
}

// 1-byte outputs with COALESCED loads: lane l of a workgroup reads float4 number base + l + u * kBlock (a wave reads 1 KiB
// contiguous per instruction, U of them in flight) and writes the 4 bytes of each float4 as one dword: 256 B contiguous per
// wave and store.  (Round 3's form -- 16 consecutive elements per lane, one 16-B store -- made every lane walk its own 64-B
// segment, four partial touches of each cache line: 0.69 of the roofline against 0.77; removed in round 5.)
// Streaming (nontemporal) loads when the tensor exceeds cache residency.
template <typename OUT, bool CHANNEL, bool NT>
__global__ __launch_bounds__(kBlock) void to_int8_vec_kernel(const float4* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ offset, uint32_t* __restrict__ out, uint32_t nvec,
                                                             FastDiv vec_per_channel, FastDiv num_channel, float qmin, float qmax,
                                                             int rounding) {
    constexpr int U = 4;
    const uint32_t base = blockIdx.x * (kBlock * U) + threadIdx.x;
    float4 a[U];
    float s[U], o[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t vv = min(base + u * kBlock, nvec - 1);      // clamped: loads stay unconditional
        a[u] = load4<NT>(&x[vv]);
        uint32_t c = 0;
        if (CHANNEL) { const uint32_t row = fdiv(vv, vec_per_channel); c = row - fdiv(row, num_channel) * num_channel.d; }
        s[u] = scale[c]; o[u] = offset[c];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t vv = base + u * kBlock;
        if (vv < nvec) {
            const int q0 = to_int_scalar(a[u].x, s[u], o[u], qmin, qmax, rounding), q1 = to_int_scalar(a[u].y, s[u], o[u], qmin, qmax, rounding);
            const int q2 = to_int_scalar(a[u].z, s[u], o[u], qmin, qmax, rounding), q3 = to_int_scalar(a[u].w, s[u], o[u], qmin, qmax, rounding);
            out[vv] = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
        }
    }
}

// --------------------------------------------------------------------------- LSQ backward
// One element of QuantizeTensor_LT_B / _LC_B (linear.cu:255-274 / :352-372).  `o` is the rounded
// offset kept as float, as in the reference; returns the partial d(loss)/d(scale) term.
// R >= 0: the rounding policy is a compile-time constant (ROUND_HALF_EVEN: one v_rndne); R < 0: runtime.
// The quotient v / s that decides the clipping mask is the IEEE division (grad_x must be exact); the
// residual term (q - v) / s * dy only feeds a float32 sum over the whole tensor, so it uses RN(1/s):
// 2 ulp on a term whose sum is order dependent to 1e-6 anyway (tests: 1e-4 relative vs the oracle).
template <bool CHANNEL, int R>
__device__ __forceinline__ float lsq_bwd_elem(float v, float dy, float s, float rcp_s, float o, int oi, int qmin,
                                              int qmax, int rounding, float* gx) {
    // (the reciprocal path of the forward kernels -- common.hpp: round_quotient4 -- was measured here in round 5 and LOST: its
    //  divergent fallback region per float4 keeps the compiler from interleaving the U load / store groups of these kernels;
    //  the weight launch of a block-wise LSQ step went 9.6 -> 17 us, Bx32 106 -> 109 us)
    const int r = (R >= 0) ? round2int_t<(R >= 0 ? R : 0)>(v / s) : round2int(v / s, rounding);
    const int qt = f2i_sat((float)r + o);
    const float q = (float)(qt - oi) * s;
    const bool hi = qt > qmax, lo = qt < qmin;
    const float inside = CHANNEL ? ((q - v) * rcp_s * dy) : ((q - v) * dy * rcp_s);
    *gx = (hi || lo) ? 0.f : dy;
    return hi ? ((float)qmax - o) * dy : (lo ? ((float)qmin - o) * dy : inside);
}

__device__ __forceinline__ float block_sum(float v, float* lds) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (wid == 0) {
        r = lane < (int)(blockDim.x >> 6) ? lds[lane] : 0.f;
        r = wave_sum(r);
    }
    return r;  // valid in wave 0
}

// per tensor: one contiguous chunk per workgroup, 16-B loads of x and dy, 16-B stores of grad_x;
// the workgroup's partial d(loss)/d(scale) goes to partial[blockIdx.x] (no same-address atomics),
// lsq_finish_kernel sums the partials in double and applies grad_factor.
#ifndef PPQHIP_LSQ_U
#define PPQHIP_LSQ_U 4                  // (x, dy) 16-B load pairs in flight per lane (MI355X sweep, Bx32: U=1 152 us, 2 127, 4 116)
#endif
#ifndef PPQHIP_LSQ_C_U
#define PPQHIP_LSQ_C_U 4
#endif
#ifndef PPQHIP_LSQ_MAX_WG
#define PPQHIP_LSQ_MAX_WG 65536         // workgroups per launch (one partial sum each)
#endif
template <int R, bool NT, int U>
__global__ __launch_bounds__(kBlock) void fq_linear_t_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    const float* __restrict__ dy, float* __restrict__ gx, float* __restrict__ partial, uint32_t n, int vec_ok,
    int qmin, int qmax, int rounding) {
    __shared__ float lds[kBlock / kWave];
    float acc = 0.f;
    uint32_t done = 0;
    const uint32_t nvec = n >> 2;
    const float4* xv = reinterpret_cast<const float4*>(x);
    const float4* dv = reinterpret_cast<const float4*>(dy);
    float4* gv = reinterpret_cast<float4*>(gx);
    const uint32_t tile = kBlock * U;
    const uint32_t tiles = (nvec + tile - 1) / tile;
    const uint32_t per = (tiles + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per * tile;
    const uint32_t hi = min(lo + per * tile, nvec);
    // the first tile's 2 U loads are issued BEFORE scale / offset are read (two dependent scalar loads and a division would
    // otherwise sit in front of them: on the 0.05 .. 6 MB activations of a block-wise LSQ step the launch is one latency chain)
    float4 a[U], d[U];
    uint32_t v = lo + threadIdx.x;
    const bool any = vec_ok && v < hi;
    if (any) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t at = min(v + u * kBlock, hi - 1);              // clamped: loads stay unconditional
            a[u] = load4<NT>(&xv[at]); d[u] = load4<NT>(&dv[at]);
        }
    }
    const float s = scale[0];
    const float rcp_s = 1.0f / s;
    const float o = __builtin_roundf(offset[0]);
    const int oi = f2i_sat(o);
    if (vec_ok) {
        while (v < hi) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (v + u * kBlock >= hi) break;
                float4 g;
                acc += lsq_bwd_elem<false, R>(a[u].x, d[u].x, s, rcp_s, o, oi, qmin, qmax, rounding, &g.x);
                acc += lsq_bwd_elem<false, R>(a[u].y, d[u].y, s, rcp_s, o, oi, qmin, qmax, rounding, &g.y);
                acc += lsq_bwd_elem<false, R>(a[u].z, d[u].z, s, rcp_s, o, oi, qmin, qmax, rounding, &g.z);
                acc += lsq_bwd_elem<false, R>(a[u].w, d[u].w, s, rcp_s, o, oi, qmin, qmax, rounding, &g.w);
                gv[v + u * kBlock] = g;
            }
            v += tile;
            if (v < hi) {                                                  // (grids are sized one tile per workgroup: rarely taken)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t at = min(v + u * kBlock, hi - 1);
                    a[u] = load4<NT>(&xv[at]); d[u] = load4<NT>(&dv[at]);
                }
            }
        }
        done = nvec << 2;
    }
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = done + blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float g;
        acc += lsq_bwd_elem<false, R>(x[i], dy[i], s, rcp_s, o, oi, qmin, qmax, rounding, &g);
        gx[i] = g;
    }
    const float tot = block_sum(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// One workgroup of 1024 lanes, 8 loads in flight per lane, double accumulation in a FIXED order (lane l adds
// partial[l], partial[l + 1024], ..; then lanes, then waves, in index order): the result does not depend on timing.
// (256 lanes adding one partial per dependent trip took 13.8 us for the 12544 partials of a [32, 512, 56, 56] tensor.)
constexpr int kLsqFinishBlock = 1024;
__global__ __launch_bounds__(kLsqFinishBlock) void lsq_finish_kernel(const float* __restrict__ partial, uint32_t count,
                                                                     float grad_factor, float* __restrict__ gs) {
    __shared__ double lds[kLsqFinishBlock / kWave];
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < count; i += 8 * kLsqFinishBlock) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t at = i + u * kLsqFinishBlock; v[u] = at < count ? partial[at] : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += (double)v[u];
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kLsqFinishBlock / kWave; w++) t += lds[w];
        gs[0] = (float)t * grad_factor;
    }
}

// the same sum for MANY scale gradients in one launch: workgroup j finishes job j exactly as lsq_finish_kernel would (same
// lanes, same order, same double accumulation -> bit-identical).  A block-wise LSQ step back-propagates through 5-10 activation
// delegators; each main kernel leaves its partials in a caller-owned buffer and ONE launch at the end of the sweep turns them
// into the scale gradients (scales are autograd leaves: nothing further back waits for them).
constexpr int kLsqFinishMax = 96;
struct LsqFinishJob { const float* partial; float* gs; uint32_t count; float grad_factor; };      // 24 B
struct LsqFinishArgs { LsqFinishJob job[kLsqFinishMax]; };
static_assert(sizeof(LsqFinishArgs) <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(kLsqFinishBlock) void lsq_finish_multi_kernel(const LsqFinishArgs args) {
    __shared__ double lds[kLsqFinishBlock / kWave];
    const LsqFinishJob& j = args.job[blockIdx.x];
    const float* __restrict__ partial = j.partial;
    const uint32_t count = j.count;
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < count; i += 8 * kLsqFinishBlock) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t at = i + u * kLsqFinishBlock; v[u] = at < count ? partial[at] : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += (double)v[u];
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kLsqFinishBlock / kWave; w++) t += lds[w];
        j.gs[0] = (float)t * j.grad_factor;
    }
}

// rows = outer * C rows of `epc` contiguous elements; block b handles chunk (b % chunks) of row
// (b / chunks) -> one channel per block, one atomic per block (spread over C addresses).
template <int R, bool NT>
__global__ __launch_bounds__(kBlock) void fq_linear_c_bwd_row_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    const float* __restrict__ dy, float* __restrict__ gx, float* __restrict__ gs, uint32_t epc, int vec_ok,
    FastDiv chunks, FastDiv num_channel, uint32_t chunk_elems, int qmin, int qmax, float grad_factor,
    int rounding) {
    __shared__ float lds[kBlock / kWave];
    const uint32_t row = fdiv(blockIdx.x, chunks);
    const uint32_t chunk = blockIdx.x - row * chunks.d;
    const uint32_t c = row - fdiv(row, num_channel) * num_channel.d;
    const float s = scale[c];
    const float rcp_s = 1.0f / s;
    const float o = __builtin_roundf(offset[c]);
    const int oi = f2i_sat(o);
    const uint32_t lo = chunk * chunk_elems;
    const uint32_t hi = min(lo + chunk_elems, epc);
    const size_t base = (size_t)row * epc;
    float acc = 0.f;
    if (vec_ok) {   // epc % 4 == 0, chunk_elems % 4 == 0, 16-B aligned bases
        const float4* xv = reinterpret_cast<const float4*>(x + base);
        const float4* dv = reinterpret_cast<const float4*>(dy + base);
        float4* gv = reinterpret_cast<float4*>(gx + base);
        // U (x, dy) load pairs in flight per lane: a 56 x 56 row (784 float4) is ONE trip of 4, all 8 loads issued up front
        // (one pair per dependent trip kept this kernel at 0.68 of the roofline on [32, 512, 56, 56])
        constexpr int U = PPQHIP_LSQ_C_U;
        const uint32_t v1 = hi >> 2;
        for (uint32_t v = (lo >> 2) + threadIdx.x; v < v1; v += kBlock * U) {
            float4 a[U], d[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t at = min(v + u * kBlock, v1 - 1);          // clamped: loads stay unconditional
                a[u] = load4<NT>(&xv[at]); d[u] = load4<NT>(&dv[at]);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (v + u * kBlock >= v1) break;
                float4 g;
                acc += lsq_bwd_elem<true, R>(a[u].x, d[u].x, s, rcp_s, o, oi, qmin, qmax, rounding, &g.x);
                acc += lsq_bwd_elem<true, R>(a[u].y, d[u].y, s, rcp_s, o, oi, qmin, qmax, rounding, &g.y);
                acc += lsq_bwd_elem<true, R>(a[u].z, d[u].z, s, rcp_s, o, oi, qmin, qmax, rounding, &g.z);
                acc += lsq_bwd_elem<true, R>(a[u].w, d[u].w, s, rcp_s, o, oi, qmin, qmax, rounding, &g.w);
                gv[v + u * kBlock] = g;
            }
        }
    } else {
        for (uint32_t j = lo + threadIdx.x; j < hi; j += kBlock) {
            float g;
            acc += lsq_bwd_elem<true, R>(x[base + j], dy[base + j], s, rcp_s, o, oi, qmin, qmax, rounding, &g);
            gx[base + j] = g;
        }
    }
    const float tot = block_sum(acc, lds);
    if (threadIdx.x == 0) atomicAdd(&gs[c], tot * grad_factor);
}

// generic layout (small elem_per_channel, e.g. channel-last): per-element atomics into LDS
// accumulators (num_channel <= lds_channels) or straight to global memory.
__global__ __launch_bounds__(kBlock) void fq_linear_c_bwd_generic_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    const float* __restrict__ dy, float* __restrict__ gx, float* __restrict__ gs, uint32_t n,
    FastDiv elem_per_channel, FastDiv num_channel, int use_lds, int qmin, int qmax, float grad_factor,
    int rounding) {
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
        float g;
        const float sc = scale[c], oc = __builtin_roundf(offset[c]);
        const float p = lsq_bwd_elem<true, -1>(x[i], dy[i], sc, 1.0f / sc, oc, f2i_sat(oc), qmin, qmax, rounding, &g);
        gx[i] = g;
        if (use_lds) atomicAdd(&acc_lds[c], p);
        else atomicAdd(&gs[c], p * grad_factor);
    }
    if (use_lds) {
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < C; c += kBlock) {
            const float v = acc_lds[c];
            if (v != 0.f) atomicAdd(&gs[c], v * grad_factor);
        }
    }
}

// --------------------------------------------------------------------------- host dispatch
static int validate_n(int64_t n, const char* what) {
    if (n <= 0) { set_error("%s: tensor is empty", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (n > 0x7fffffffLL) {
        set_error("%s: there are too many elements in your tensor (more than 2*10^9)", what);
        return PPQHIP_ERR_INVALID_VALUE;
    }
    return PPQHIP_OK;
}

static int validate_channels(int64_t n, int64_t C, int64_t epc, const char* what) {
    if (C <= 0 || epc <= 0 || n % (C * epc) != 0) {
        set_error("%s: n=%lld is not [outer, %lld channels, %lld elem/channel]", what, (long long)n,
                  (long long)C, (long long)epc);
        return PPQHIP_ERR_INVALID_VALUE;
    }
    return PPQHIP_OK;
}

// Tensors of at least kStreamElems elements (192 MiB) cannot be resident in the 256 MiB Infinity
// Cache together with their output: read them with streaming (nontemporal) loads.
constexpr int64_t kStreamElems = 48ll << 20;
// Tile shape: U 16-B accesses per lane and workgroup.  HBM-bound tensors stream best with U = 2 (half as many workgroups to
// dispatch); a tensor of a few MB is ONE latency chain (dispatch -> loads -> arithmetic -> stores) in which the arithmetic of
// the last-arriving wave is exposed, so U = 1 (twice the waves, half the work behind each load) is 0.2 us faster on
// B = [1,512,56,56]: 4.28 -> 4.08 us per tensor, 4.44 -> 4.24 per channel (rocprofv3 medians, interleaved A/B,
// profiles/r05_floor_table.txt); the copy floor of the same 12.8 MB is 3.88 us.
#ifndef PPQHIP_FQ_U
#define PPQHIP_FQ_U 2
#endif
#ifndef PPQHIP_FQ_SMALL_U
#define PPQHIP_FQ_SMALL_U 1
#endif
#ifndef PPQHIP_FQ_SMALL_ELEMS
#define PPQHIP_FQ_SMALL_ELEMS (4ll << 20)      // up to 16 MB in + 16 MB out
#endif
constexpr int kTileU = PPQHIP_FQ_U;
constexpr int kSmallU = PPQHIP_FQ_SMALL_U;

template <int R>
static void launch_lt(const float* x, const float* scale, const float* offset, float* out, int64_t n,
                      int qmin, int qmax, int rounding, hipStream_t st) {
    if (aligned16(x) && aligned16(out) && n >= 4) {
        const uint32_t nvec = (uint32_t)(n >> 2);
        const int ntail = (int)(n & 3);
        const float* xt = x + (size_t)nvec * 4;
        float* ot = out + (size_t)nvec * 4;
#define PPQ_LAUNCH_LT(U, NT)                                                                                          \
        hipLaunchKernelGGL((fq_linear_t_tile_kernel<R, U, NT>), dim3((nvec + kBlock * U - 1) / (kBlock * U)), dim3(kBlock), 0, st, \
                           (const float4*)x, scale, offset, (float4*)out, nvec, xt, ot, ntail, qmin, qmax, rounding)
        if (n >= kStreamElems) PPQ_LAUNCH_LT(kTileU, true);
        else if (n <= PPQHIP_FQ_SMALL_ELEMS) PPQ_LAUNCH_LT(kSmallU, false);
        else PPQ_LAUNCH_LT(kTileU, false);
#undef PPQ_LAUNCH_LT
    } else {
        hipLaunchKernelGGL((fq_linear_t_scalar_kernel<R>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, st, x,
                           scale, offset, out, (uint32_t)n, qmin, qmax, rounding);
    }
}

template <int R>
static void launch_lc(const float* x, const float* scale, const float* offset, float* out, int64_t n,
                      int64_t C, int64_t epc, int qmin, int qmax, int rounding, hipStream_t st) {
    if (aligned16(x) && aligned16(out) && (epc % 4 == 0)) {
        const uint32_t nvec = (uint32_t)(n >> 2);
        const FastDiv vpc = make_fastdiv((uint32_t)(epc / 4)), nc = make_fastdiv((uint32_t)C);
#define PPQ_LAUNCH_LC(U, NT)                                                                                          \
        hipLaunchKernelGGL((fq_linear_c_tile_kernel<R, U, NT>), dim3((nvec + kBlock * U - 1) / (kBlock * U)), dim3(kBlock), 0, st, \
                           (const float4*)x, scale, offset, (float4*)out, nvec, vpc, nc, qmin, qmax, rounding)
        if (n >= kStreamElems) PPQ_LAUNCH_LC(kTileU, true);
        else if (n <= PPQHIP_FQ_SMALL_ELEMS) PPQ_LAUNCH_LC(kSmallU, false);
        else PPQ_LAUNCH_LC(kTileU, false);
#undef PPQ_LAUNCH_LC
    } else {
        const FastDiv e = make_fastdiv((uint32_t)epc), nc = make_fastdiv((uint32_t)C);
        hipLaunchKernelGGL((fq_linear_c_scalar_kernel<R>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, st, x,
                           scale, offset, out, (uint32_t)n, e, nc, qmin, qmax, rounding);
    }
}

// ---- many tensors, one launch --------------------------------------------------------------------
// Every forward of a quantised graph fake-quantises all of its weights again (the reference executor
// does so until ParameterBakingPass); ResNet-50 has 54 of them, 0.01 .. 9 MB each -- one launch per
// weight is pure launch latency.  Here one launch serves them all: the job table lives in device
// memory (it only changes when a weight / scale tensor is replaced), the kernel arguments carry the
// prefix of workgroup counts so a workgroup finds its job without touching memory, and each
// workgroup quantises one tile of kBlock * U float4 (or the same range element-wise when the tensor
// cannot be vectorised: elem_per_channel % 4 != 0 or unaligned).  Per-tensor jobs are C = 1.
constexpr int kFqMultiMax = 128;
struct FqJob {
    const float* x;
    float* out;
    const float* scale;
    const float* offset;
    uint32_t n;
    uint32_t vec_ok;
    FastDiv per;          // vec_ok: float4 per channel row; else elements per channel row
    FastDiv nc;
    int qmin, qmax;
};
struct FqMultiArgs {
    uint32_t first_block[kFqMultiMax];
    uint32_t count;
    int rounding;
    const FqJob* jobs;
};

template <int R, int U>
__global__ __launch_bounds__(kBlock) void fq_linear_multi_kernel(const FqMultiArgs args) {
    uint32_t lo = 0, hi = args.count;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (args.first_block[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    const FqJob j = args.jobs[lo];                      // uniform address: scalar loads
    const uint32_t base = (blockIdx.x - args.first_block[lo]) * (kBlock * U) + threadIdx.x;
    const uint32_t C = j.nc.d;
    if (j.vec_ok) {
        const uint32_t nvec = j.n >> 2;
        const float4* xv = reinterpret_cast<const float4*>(j.x);
        float4* ov = reinterpret_cast<float4*>(j.out);
        float4 a[U];
        float s[U];
        int o[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const uint32_t vv = min(base + k * kBlock, nvec - 1);
            a[k] = xv[vv];
            const uint32_t row = fdiv(vv, j.per);
            const uint32_t c = row - fdiv(row, j.nc) * C;
            s[k] = j.scale[c];
            o[k] = round_offset(j.offset[c]);
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            const uint32_t vv = base + k * kBlock;
            const float4 r = fq_linear4<R>(a[k], s[k], fq_safe_rcp(s[k]), o[k], j.qmin, j.qmax, args.rounding);
            if (vv < nvec) ov[vv] = r;
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
                    j.out[e] = fq_linear_scalar<R>(j.x[e], j.scale[c], round_offset(j.offset[c]), j.qmin, j.qmax,
                                                   args.rounding);
                }
            }
        }
    }
}

// ---- LSQ backward of many per-channel tensors, one launch ------------------------------------------
// A block-wise LSQ step (LearnedStepSizePass, optim/training.py:728-826) back-propagates through every weight of the
// block: 5-10 tensors of 0.01 .. 10 MB, each a memset + a kernel of ~6 us in the per-tensor path (BENCH_r03: 224 launches
// of 6.4 us on 3.7 MB = 0.07 of the roofline).  Here ONE launch serves them all: workgroup b -> (job, channel c); it
// walks the `outer` rows of its channel with the row kernel's loop (U 16-B (x, dy) pairs in flight), block-reduces and
// STORES grad_s[c] -- one workgroup owns a channel, so there is no atomic, no memset and the sum order is fixed.
// With outer == 1 and 256 <= elem_per_channel <= 4096 (convolution weights, channel axis 0) the order is exactly the
// per-tensor row kernel's, so grad_s is bit-identical to ppqhip_fq_linear_c_bwd; grad_x always is.
// The job table travels BY VALUE in the kernel arguments (<= 32 jobs, 2.7 KB): dy is a fresh autograd tensor every step,
// so a device-resident table would need an upload per step; by value it costs nothing and is graph-capturable.
constexpr int kLsqMultiMax = 32;
struct LsqJob {
    const float* x;
    const float* dy;
    float* gx;
    float* gs;
    const float* scale;
    const float* offset;
    uint32_t C, epc, outer, vec_ok;
    int qmin, qmax;
    float grad_factor;
    uint32_t pad;
};
struct LsqMultiArgs {
    LsqJob jobs[kLsqMultiMax];
    uint32_t first_block[kLsqMultiMax];
    uint32_t count;
    int rounding;
};

template <int R>
__global__ __launch_bounds__(kBlock) void fq_linear_c_bwd_multi_kernel(const LsqMultiArgs args) {
    __shared__ float lds[kBlock / kWave];
    uint32_t lo = 0, hi = args.count;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (args.first_block[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    const LsqJob& j = args.jobs[lo];                    // kernel-argument segment: scalar loads
    const uint32_t c = blockIdx.x - args.first_block[lo];
    const float s = j.scale[c];
    const float rcp_s = 1.0f / s;
    const float o = __builtin_roundf(j.offset[c]);
    const int oi = f2i_sat(o);
    const uint32_t epc = j.epc, C = j.C;
    const int qmin = j.qmin, qmax = j.qmax, rounding = args.rounding;
    float acc = 0.f;
    for (uint32_t n = 0; n < j.outer; n++) {
        const size_t base = ((size_t)n * C + c) * epc;
        if (j.vec_ok) {   // epc % 4 == 0, 16-B aligned bases
            const float4* xv = reinterpret_cast<const float4*>(j.x + base);
            const float4* dv = reinterpret_cast<const float4*>(j.dy + base);
            float4* gv = reinterpret_cast<float4*>(j.gx + base);
            constexpr int U = PPQHIP_LSQ_C_U;
            const uint32_t v1 = epc >> 2;
            for (uint32_t v = threadIdx.x; v < v1; v += kBlock * U) {
                float4 a[U], d[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t at = min(v + u * kBlock, v1 - 1);
                    a[u] = xv[at]; d[u] = dv[at];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (v + u * kBlock >= v1) break;
                    float4 g;
                    acc += lsq_bwd_elem<true, R>(a[u].x, d[u].x, s, rcp_s, o, oi, qmin, qmax, rounding, &g.x);
                    acc += lsq_bwd_elem<true, R>(a[u].y, d[u].y, s, rcp_s, o, oi, qmin, qmax, rounding, &g.y);
                    acc += lsq_bwd_elem<true, R>(a[u].z, d[u].z, s, rcp_s, o, oi, qmin, qmax, rounding, &g.z);
                    acc += lsq_bwd_elem<true, R>(a[u].w, d[u].w, s, rcp_s, o, oi, qmin, qmax, rounding, &g.w);
                    gv[v + u * kBlock] = g;
                }
            }
        } else {
            for (uint32_t e = threadIdx.x; e < epc; e += kBlock) {
                float g;
                acc += lsq_bwd_elem<true, R>(j.x[base + e], j.dy[base + e], s, rcp_s, o, oi, qmin, qmax, rounding, &g);
                j.gx[base + e] = g;
            }
        }
    }
    const float tot = block_sum(acc, lds);
    if (threadIdx.x == 0) j.gs[c] = tot * j.grad_factor;
}

}  // namespace ppqhip

using namespace ppqhip;

extern "C" {

int ppqhip_fq_linear_t(const float* x, const float* scale, const float* offset, float* out, int64_t n,
                       int clip_min, int clip_max, int rounding, void* stream) {
    if (int st = validate_n(n, "fq_linear_t")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_FQ_LINEAR_T, 8.0 * (double)n, s);
    if (rounding == ROUND_HALF_EVEN) launch_lt<ROUND_HALF_EVEN>(x, scale, offset, out, n, clip_min, clip_max, rounding, s);
    else launch_lt<-1>(x, scale, offset, out, n, clip_min, clip_max, rounding, s);
    return finish_launch("fq_linear_t");
}

int ppqhip_fq_linear_c(const float* x, const float* scale, const float* offset, float* out, int64_t n,
                       int64_t num_channel, int64_t elem_per_channel, int clip_min, int clip_max,
                       int rounding, void* stream) {
    if (int st = validate_n(n, "fq_linear_c")) return st;
    if (int st = validate_channels(n, num_channel, elem_per_channel, "fq_linear_c")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_FQ_LINEAR_C, 8.0 * (double)n, s);
    if (rounding == ROUND_HALF_EVEN)
        launch_lc<ROUND_HALF_EVEN>(x, scale, offset, out, n, num_channel, elem_per_channel, clip_min, clip_max,
                                   rounding, s);
    else launch_lc<-1>(x, scale, offset, out, n, num_channel, elem_per_channel, clip_min, clip_max, rounding, s);
    return finish_launch("fq_linear_c");
}

int64_t ppqhip_fq_linear_multi_table_bytes(int num_jobs) {
    return num_jobs > 0 ? (int64_t)sizeof(FqJob) * num_jobs : 0;
}

int ppqhip_fq_linear_multi(const ppqhip_fq_job* jobs, int num_jobs, int rounding, void* device_table, int upload,
                           void* stream) {
    if (num_jobs <= 0) return PPQHIP_OK;
    if (jobs == nullptr || device_table == nullptr) {
        set_error("fq_linear_multi: jobs / device_table is null"); return PPQHIP_ERR_INVALID_VALUE;
    }
    hipStream_t s = (hipStream_t)stream;
    double bytes = 0.0;
    for (int k = 0; k < num_jobs; k++) {
        if (int st = validate_n(jobs[k].n, "fq_linear_multi")) return st;
        if (int st = validate_channels(jobs[k].n, jobs[k].num_channel, jobs[k].elem_per_channel, "fq_linear_multi")) return st;
        if (!jobs[k].x || !jobs[k].out || !jobs[k].scale || !jobs[k].offset) {
            set_error("fq_linear_multi: job %d has a null pointer", k); return PPQHIP_ERR_INVALID_VALUE;
        }
        bytes += 8.0 * (double)jobs[k].n;
    }
    LaunchScope scope(K_FQ_LINEAR_C, bytes, s);
#ifndef PPQHIP_FQM_U
#define PPQHIP_FQM_U 2
#endif
    constexpr int U = PPQHIP_FQM_U;
    for (int base = 0; base < num_jobs; base += kFqMultiMax) {
        const int count = (num_jobs - base) < kFqMultiMax ? (num_jobs - base) : kFqMultiMax;
        FqMultiArgs args;
        args.count = (uint32_t)count; args.rounding = rounding;
        args.jobs = (const FqJob*)device_table + base;
        std::vector<FqJob> table(upload ? count : 0);
        uint32_t blocks = 0;
        for (int k = 0; k < count; k++) {
            const ppqhip_fq_job& src = jobs[base + k];
            const bool vec = aligned16(src.x) && aligned16(src.out) && (src.elem_per_channel % 4 == 0);
            args.first_block[k] = blocks;
            const uint64_t quads = ((uint64_t)src.n + 3) / 4;
            blocks += (uint32_t)((quads + kBlock * U - 1) / (kBlock * U));
            if (upload) {
                FqJob& d = table[k];
                d.x = src.x; d.out = src.out; d.scale = src.scale; d.offset = src.offset;
                d.n = (uint32_t)src.n; d.vec_ok = vec ? 1u : 0u;
                d.per = make_fastdiv((uint32_t)(vec ? src.elem_per_channel / 4 : src.elem_per_channel));
                d.nc = make_fastdiv((uint32_t)src.num_channel);
                d.qmin = src.clip_min; d.qmax = src.clip_max;
            }
        }
        if (upload) {
            // pageable source: the runtime stages the copy before returning, `table` may go out of scope
            if (int st = check_hip(hipMemcpyAsync((FqJob*)device_table + base, table.data(), sizeof(FqJob) * count,
                                                  hipMemcpyHostToDevice, s), "fq_linear_multi table upload"))
                return st;
        }
        if (rounding == ROUND_HALF_EVEN)
            hipLaunchKernelGGL((fq_linear_multi_kernel<ROUND_HALF_EVEN, U>), dim3(blocks), dim3(kBlock), 0, s, args);
        else
            hipLaunchKernelGGL((fq_linear_multi_kernel<-1, U>), dim3(blocks), dim3(kBlock), 0, s, args);
    }
    return finish_launch("fq_linear_multi");
}

static int to_int_impl(const float* x, const float* scale, const float* offset, void* out, int64_t n, int64_t C, int64_t epc,
                       int clip_min, int clip_max, int rounding, int out_dtype, bool channel, void* stream, const char* what) {
    if (int st = validate_n(n, what)) return st;
    if (channel) { if (int st = validate_channels(n, C, epc, what)) return st; }
    if (out_dtype < 0 || out_dtype > 2) { set_error("%s: out_dtype must be 0 (int8), 1 (uint8) or 2 (int32)", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (rounding == ROUND_TO_NEAR_INT || rounding < 0 || rounding > ROUND_UP) {
        // utils/round.py:47-49: the tensor form of this policy does not exist in the reference either
        set_error("%s: rounding policy %d has no tensor form (ppq_tensor_round)", what, rounding); return PPQHIP_ERR_INVALID_VALUE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int elem = out_dtype == 2 ? 4 : 1;
    const int vec_ok = (aligned16(x) && (reinterpret_cast<uintptr_t>(out) % (size_t)(4 * elem) == 0) && (!channel || epc % 4 == 0)) ? 1 : 0;
    const FastDiv e = make_fastdiv((uint32_t)(channel ? epc : 1)), nc = make_fastdiv((uint32_t)(channel ? C : 1));
    const dim3 grid((uint32_t)((n + kBlock * 4 - 1) / (kBlock * 4)));
    const float qmin = (float)clip_min, qmax = (float)clip_max;
    // 1-byte outputs, 16-B aligned input, 4-B aligned output, channels in whole float4s: the coalesced kernel takes n / 4 groups,
    // the (< 4 element) rest goes through the general one
    if (out_dtype != 2 && aligned16(x) && (reinterpret_cast<uintptr_t>(out) % 4 == 0) && (!channel || epc % 4 == 0) && n >= 4) {
        const uint32_t nvec = (uint32_t)(n / 4);
        const dim3 gv((nvec + kBlock * 4 - 1) / (kBlock * 4));
        const FastDiv vpc = make_fastdiv((uint32_t)(channel ? epc / 4 : 1));
        const bool nt = n >= kStreamElems;
#define PPQ_LAUNCH_TOINTV(T, CH, NT) hipLaunchKernelGGL((to_int8_vec_kernel<T, CH, NT>), gv, dim3(kBlock), 0, s, (const float4*)x, scale, \
                                                        offset, (uint32_t*)out, nvec, vpc, nc, qmin, qmax, rounding)
        if (out_dtype == 0) {
            if (channel) { if (nt) PPQ_LAUNCH_TOINTV(int8_t, true, true); else PPQ_LAUNCH_TOINTV(int8_t, true, false); }
            else { if (nt) PPQ_LAUNCH_TOINTV(int8_t, false, true); else PPQ_LAUNCH_TOINTV(int8_t, false, false); }
        } else {
            if (channel) { if (nt) PPQ_LAUNCH_TOINTV(uint8_t, true, true); else PPQ_LAUNCH_TOINTV(uint8_t, true, false); }
            else { if (nt) PPQ_LAUNCH_TOINTV(uint8_t, false, true); else PPQ_LAUNCH_TOINTV(uint8_t, false, false); }
        }
#undef PPQ_LAUNCH_TOINTV
        const int64_t done = (int64_t)nvec * 4;
        if (done == n) return finish_launch(what);
        if (!channel) return to_int_impl(x + done, scale, offset, (uint8_t*)out + done, n - done, 1, n - done, clip_min, clip_max, rounding,
                                         out_dtype, false, stream, what);
        set_error("%s: internal: ragged tail with channels", what); return PPQHIP_ERR_INVALID_VALUE;      // n % (C * epc) == 0 and epc % 4 == 0
    }
#define PPQ_LAUNCH_TOINT(T, CH) hipLaunchKernelGGL((to_int_kernel<T, CH>), grid, dim3(kBlock), 0, s, x, scale, offset, (T*)out, \
                                                   (uint32_t)n, vec_ok, e, nc, qmin, qmax, rounding)
    if (out_dtype == 0) { if (channel) PPQ_LAUNCH_TOINT(int8_t, true); else PPQ_LAUNCH_TOINT(int8_t, false); }
    else if (out_dtype == 1) { if (channel) PPQ_LAUNCH_TOINT(uint8_t, true); else PPQ_LAUNCH_TOINT(uint8_t, false); }
    else { if (channel) PPQ_LAUNCH_TOINT(int32_t, true); else PPQ_LAUNCH_TOINT(int32_t, false); }
#undef PPQ_LAUNCH_TOINT
    return finish_launch(what);
}

int ppqhip_to_int_t(const float* x, const float* scale, const float* offset, void* out, int64_t n, int clip_min, int clip_max,
                    int rounding, int out_dtype, void* stream) {
    return to_int_impl(x, scale, offset, out, n, 1, n, clip_min, clip_max, rounding, out_dtype, false, stream, "to_int_t");
}

int ppqhip_to_int_c(const float* x, const float* scale, const float* offset, void* out, int64_t n, int64_t num_channel,
                    int64_t elem_per_channel, int clip_min, int clip_max, int rounding, int out_dtype, void* stream) {
    return to_int_impl(x, scale, offset, out, n, num_channel, elem_per_channel, clip_min, clip_max, rounding, out_dtype, true, stream,
                       "to_int_c");
}

// (x, dy) load pairs per lane and workgroup: 4 for HBM-bound tensors (sweep in profiles/r03_lsq_variants.txt), 1 for the
// latency-bound ones (up to 16 MB per operand: four times the waves, a quarter of the arithmetic behind each load -- the
// activations of a block-wise LSQ step are 0.05 .. 6 MB)
#ifndef PPQHIP_LSQ_SMALL_U
#define PPQHIP_LSQ_SMALL_U 1
#endif
static int lsq_t_u(int64_t n) { return n <= (4ll << 20) ? PPQHIP_LSQ_SMALL_U : PPQHIP_LSQ_U; }
static int lsq_t_grid(int64_t n) { return stream_grid(n, kBlock * 4 * lsq_t_u(n), PPQHIP_LSQ_MAX_WG); }

static void launch_lsq_t_main(const float* x, const float* scale, const float* offset, const float* grad_y, float* grad_x,
                              float* partial, int64_t n, int grid, int clip_min, int clip_max, int rounding, hipStream_t s) {
    const int vec_ok = (aligned16(x) && aligned16(grad_y) && aligned16(grad_x)) ? 1 : 0;
    const bool nt = n >= kStreamElems / 2;       // x and dy together exceed cache residency: streaming loads
#define PPQ_LAUNCH_LSQ_T(R, NT, U)                                                                                  \
    hipLaunchKernelGGL((fq_linear_t_bwd_kernel<R, NT, U>), dim3(grid), dim3(kBlock), 0, s, x, scale, offset, grad_y, \
                       grad_x, partial, (uint32_t)n, vec_ok, clip_min, clip_max, rounding)
    if (lsq_t_u(n) == PPQHIP_LSQ_SMALL_U) {          // never nontemporal: these tensors are cache resident
        if (rounding == ROUND_HALF_EVEN) PPQ_LAUNCH_LSQ_T(ROUND_HALF_EVEN, false, PPQHIP_LSQ_SMALL_U); else PPQ_LAUNCH_LSQ_T(-1, false, PPQHIP_LSQ_SMALL_U);
    } else if (rounding == ROUND_HALF_EVEN) { if (nt) PPQ_LAUNCH_LSQ_T(ROUND_HALF_EVEN, true, PPQHIP_LSQ_U); else PPQ_LAUNCH_LSQ_T(ROUND_HALF_EVEN, false, PPQHIP_LSQ_U); }
    else { if (nt) PPQ_LAUNCH_LSQ_T(-1, true, PPQHIP_LSQ_U); else PPQ_LAUNCH_LSQ_T(-1, false, PPQHIP_LSQ_U); }
#undef PPQ_LAUNCH_LSQ_T
}

int64_t ppqhip_fq_linear_t_bwd_partials(int64_t n) { return n > 0 && n <= 0x7fffffffLL ? (int64_t)lsq_t_grid(n) : 0; }

int ppqhip_fq_linear_t_bwd_main(const float* x, const float* scale, const float* offset, const float* grad_y, float* grad_x,
                                float* partial, int64_t n, int clip_min, int clip_max, int rounding, void* stream) {
    if (int st = validate_n(n, "fq_linear_t_bwd_main")) return st;
    if (partial == nullptr) { set_error("fq_linear_t_bwd_main: partial is null"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_FQ_LINEAR_T_BWD, 12.0 * (double)n, s);
    launch_lsq_t_main(x, scale, offset, grad_y, grad_x, partial, n, lsq_t_grid(n), clip_min, clip_max, rounding, s);
    return finish_launch("fq_linear_t_bwd_main");
}

int ppqhip_lsq_finish_multi(const ppqhip_lsq_finish_job* jobs, int num_jobs, void* stream) {
    if (num_jobs <= 0) return PPQHIP_OK;
    if (jobs == nullptr) { set_error("lsq_finish_multi: jobs is null"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    double bytes = 0.0;
    for (int k = 0; k < num_jobs; k++) if (jobs[k].n > 0) bytes += 4.0 * (double)lsq_t_grid(jobs[k].n);      // the partial sums it folds
    LaunchScope scope(K_LSQ_FINISH, bytes, s);
    for (int base = 0; base < num_jobs; base += kLsqFinishMax) {
        const int count = (num_jobs - base) < kLsqFinishMax ? (num_jobs - base) : kLsqFinishMax;
        LsqFinishArgs args;
        for (int k = 0; k < count; k++) {
            const ppqhip_lsq_finish_job& src = jobs[base + k];
            if (int st = validate_n(src.n, "lsq_finish_multi")) return st;
            if (!src.partial || !src.grad_s) { set_error("lsq_finish_multi: job %d has a null pointer", base + k); return PPQHIP_ERR_INVALID_VALUE; }
            args.job[k].partial = src.partial; args.job[k].gs = src.grad_s;
            args.job[k].count = (uint32_t)lsq_t_grid(src.n);
            // rsqrtf(((double)n * (clip_max - clip_min))): linear.cu:299
            args.job[k].grad_factor = (float)(1.0 / sqrt((double)src.n * (double)(src.clip_max - src.clip_min)));
        }
        for (int k = count; k < kLsqFinishMax; k++) args.job[k] = args.job[0];
        hipLaunchKernelGGL(lsq_finish_multi_kernel, dim3(count), dim3(kLsqFinishBlock), 0, s, args);
    }
    return finish_launch("lsq_finish_multi");
}

int ppqhip_fq_linear_t_bwd(const float* x, const float* scale, const float* offset, const float* grad_y,
                           float* grad_x, float* grad_s, int64_t n, int clip_min, int clip_max,
                           int rounding, void* stream) {
    if (int st = validate_n(n, "fq_linear_t_bwd")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_FQ_LINEAR_T_BWD, 12.0 * (double)n, s);
    // rsqrtf(((double)n * (clip_max - clip_min))): linear.cu:299
    const float grad_factor = (float)(1.0 / sqrt((double)n * (double)(clip_max - clip_min)));
    // one tile per workgroup while that keeps the grid below PPQHIP_LSQ_MAX_WG: tens of thousands of short
    // workgroups stream better than a chip-sized persistent grid for this 2-reads + 1-write pattern (same finding as
    // the forward tile kernels).  Measured again in round 3 with ping-pong register tiles and the partial sum folded into
    // the launch (last workgroup, sharded tickets): 512 x 2 / CU 120 us, 512 x 4 122, 256 x 8 125, U = 4 124 -- against
    // 107 us for this kernel + its finish launch (profiles/r03_lsq_variants.txt)
    const int grid = lsq_t_grid(n);
    float* partial = (float*)scratch(s, sizeof(float) * (size_t)grid);
    if (partial == nullptr) return PPQHIP_ERR_HIP;
    launch_lsq_t_main(x, scale, offset, grad_y, grad_x, partial, n, grid, clip_min, clip_max, rounding, s);
    hipLaunchKernelGGL(lsq_finish_kernel, dim3(1), dim3(kLsqFinishBlock), 0, s, (const float*)partial, (uint32_t)grid,
                       grad_factor, grad_s);
    return finish_launch("fq_linear_t_bwd");
}

int ppqhip_fq_linear_c_bwd(const float* x, const float* scale, const float* offset, const float* grad_y,
                           float* grad_x, float* grad_s, int64_t n, int64_t num_channel,
                           int64_t elem_per_channel, int clip_min, int clip_max, int rounding,
                           void* stream) {
    if (int st = validate_n(n, "fq_linear_c_bwd")) return st;
    if (int st = validate_channels(n, num_channel, elem_per_channel, "fq_linear_c_bwd")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_FQ_LINEAR_C_BWD, 12.0 * (double)n, s);
    if (int st = check_hip(hipMemsetAsync(grad_s, 0, sizeof(float) * (size_t)num_channel, s), "memset grad_s"))
        return st;
    // rsqrtf(((double)n * clip_max)): linear.cu:402
    const float grad_factor = (float)(1.0 / sqrt((double)n * (double)clip_max));
    const FastDiv nc = make_fastdiv((uint32_t)num_channel);
    if (elem_per_channel >= 256) {
        const uint32_t chunk_elems = 4096;
        const uint32_t chunks = (uint32_t)((elem_per_channel + chunk_elems - 1) / chunk_elems);
        const int64_t rows = n / elem_per_channel;
        const int vec_ok = (elem_per_channel % 4 == 0 && aligned16(x) && aligned16(grad_y) && aligned16(grad_x)) ? 1 : 0;
        const bool nt = n >= kStreamElems / 2;      // x and dy together exceed cache residency: streaming loads (as lsq_bwd_t)
#define PPQ_LAUNCH_LSQ_C(R, NT)                                                                                        \
        hipLaunchKernelGGL((fq_linear_c_bwd_row_kernel<R, NT>), dim3((uint32_t)(rows * chunks)), dim3(kBlock), 0, s, x,   \
                           scale, offset, grad_y, grad_x, grad_s, (uint32_t)elem_per_channel, vec_ok,                  \
                           make_fastdiv(chunks), nc, chunk_elems, clip_min, clip_max, grad_factor, rounding)
        if (rounding == ROUND_HALF_EVEN) { if (nt) PPQ_LAUNCH_LSQ_C(ROUND_HALF_EVEN, true); else PPQ_LAUNCH_LSQ_C(ROUND_HALF_EVEN, false); }
        else { if (nt) PPQ_LAUNCH_LSQ_C(-1, true); else PPQ_LAUNCH_LSQ_C(-1, false); }
#undef PPQ_LAUNCH_LSQ_C
    } else {
        const int use_lds = num_channel <= 8192;
        const size_t lds = use_lds ? sizeof(float) * (size_t)num_channel : 0;
        hipLaunchKernelGGL(fq_linear_c_bwd_generic_kernel, dim3(stream_grid(n, kBlock * 8, num_cu() * 2)),
                           dim3(kBlock), lds, s, x, scale, offset, grad_y, grad_x, grad_s, (uint32_t)n,
                           make_fastdiv((uint32_t)elem_per_channel), nc, use_lds, clip_min, clip_max,
                           grad_factor, rounding);
    }
    return finish_launch("fq_linear_c_bwd");
}

int ppqhip_fq_linear_c_bwd_multi(const ppqhip_lsq_job* jobs, int num_jobs, int rounding, void* stream) {
    if (num_jobs <= 0) return PPQHIP_OK;
    if (jobs == nullptr) { set_error("fq_linear_c_bwd_multi: jobs is null"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    double bytes = 0.0;
    for (int k = 0; k < num_jobs; k++) {
        const ppqhip_lsq_job& j = jobs[k];
        if (int st = validate_n(j.n, "fq_linear_c_bwd_multi")) return st;
        if (int st = validate_channels(j.n, j.num_channel, j.elem_per_channel, "fq_linear_c_bwd_multi")) return st;
        if (!j.x || !j.scale || !j.offset || !j.grad_y || !j.grad_x || !j.grad_s) {
            set_error("fq_linear_c_bwd_multi: job %d has a null pointer", k); return PPQHIP_ERR_INVALID_VALUE;
        }
        bytes += 12.0 * (double)j.n;
    }
    LaunchScope scope(K_FQ_LINEAR_C_BWD, bytes, s);
    for (int base = 0; base < num_jobs; base += kLsqMultiMax) {
        const int count = (num_jobs - base) < kLsqMultiMax ? (num_jobs - base) : kLsqMultiMax;
        LsqMultiArgs args;
        args.count = (uint32_t)count; args.rounding = rounding;
        uint32_t blocks = 0;
        for (int k = 0; k < count; k++) {
            const ppqhip_lsq_job& src = jobs[base + k];
            LsqJob& d = args.jobs[k];
            d.x = src.x; d.dy = src.grad_y; d.gx = src.grad_x; d.gs = src.grad_s; d.scale = src.scale; d.offset = src.offset;
            d.C = (uint32_t)src.num_channel; d.epc = (uint32_t)src.elem_per_channel;
            d.outer = (uint32_t)(src.n / (src.num_channel * src.elem_per_channel));
            d.vec_ok = (src.elem_per_channel % 4 == 0 && aligned16(src.x) && aligned16(src.grad_y) && aligned16(src.grad_x)) ? 1u : 0u;
            d.qmin = src.clip_min; d.qmax = src.clip_max;
            d.grad_factor = (float)(1.0 / sqrt((double)src.n * (double)src.clip_max));      // rsqrtf(((double)n * clip_max)): linear.cu:402
            d.pad = 0;
            args.first_block[k] = blocks;
            blocks += d.C;
        }
        for (int k = count; k < kLsqMultiMax; k++) args.first_block[k] = blocks;
        if (rounding == ROUND_HALF_EVEN)
            hipLaunchKernelGGL((fq_linear_c_bwd_multi_kernel<ROUND_HALF_EVEN>), dim3(blocks), dim3(kBlock), 0, s, args);
        else
            hipLaunchKernelGGL((fq_linear_c_bwd_multi_kernel<-1>), dim3(blocks), dim3(kBlock), 0, s, args);
    }
    return finish_launch("fq_linear_c_bwd_multi");
}

}  // extern "C"
