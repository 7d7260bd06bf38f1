// train.hip -- TensorClip / RoundingLoss helpers (replaces ppq/csrc/cuda/train.cu).
// Exported by the reference (export.cc:24-30) but never called from ppq/ (SURVEY 2.1): kept for
// ABI completeness, plain grid-stride kernels, no tuning.
#include <cmath>

#include "common.hpp"

namespace ppqhip {

__global__ __launch_bounds__(kBlock) void tensor_clip_kernel(const float* __restrict__ v, const float* __restrict__ ref,
                                                             const float* __restrict__ limit, float* __restrict__ out,
                                                             uint32_t n, FastDiv elem_per_channel, FastDiv num_channel,
                                                             int per_channel) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t c = 0;
        if (per_channel) {
            const uint32_t row = fdiv(i, elem_per_channel);
            c = row - fdiv(row, num_channel) * num_channel.d;
        }
        const float l = limit[c], lo = ref[i] - l, hi = ref[i] + l, x = v[i];
        out[i] = x > hi ? hi : (x < lo ? lo : x);   // CLIP<float>, common.cuh:70-76
    }
}

// _RoundingLoss_LT / _LC, train.cu:115-141 / :220-247.  The kernels round the offset with
// nearbyint (not round) and compare against s * (clip - o) with the int offset (LT) or the raw
// float offset (LC).
__device__ __forceinline__ float rl_elem(float v, float s, float oraw, int per_channel, int qmin, int qmax,
                                         int rounding, float* dq_out) {
    const int o = f2i_sat(__builtin_rintf(oraw));
    const float dq = fq_linear_scalar<-1>(v, s, o, qmin, qmax, rounding);
    *dq_out = dq;
    const float ofs = per_channel ? oraw : (float)o;
    float diff = __builtin_fabsf(dq - v);
    if (v > s * ((float)qmax - ofs)) diff = -1.f;   // clipped marker
    if (v < s * ((float)qmin - ofs)) diff = -1.f;
    return diff;
}

__global__ __launch_bounds__(kBlock) void rounding_loss_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                               const float* __restrict__ offset, float* __restrict__ out,
                                                               uint32_t n, FastDiv elem_per_channel, FastDiv num_channel,
                                                               int per_channel, int qmin, int qmax, int rounding,
                                                               float inv_root) {
    __shared__ float lds[kBlock / kWave];
    float acc = 0.f;
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t c = 0;
        if (per_channel) {
            const uint32_t row = fdiv(i, elem_per_channel);
            c = row - fdiv(row, num_channel) * num_channel.d;
        }
        float dq;
        const float d = rl_elem(x[i], scale[c], offset[c], per_channel, qmin, qmax, rounding, &dq);
        acc += d < 0.f ? 0.f : d;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < kBlock / kWave; w++) t += lds[w];
        atomicAdd(out, t * inv_root);
    }
}

__global__ __launch_bounds__(kBlock) void rounding_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ offset, float* __restrict__ dx,
                                                                   uint32_t n, FastDiv elem_per_channel,
                                                                   FastDiv num_channel, int per_channel, int qmin,
                                                                   int qmax, int rounding, float root) {
    const float g = dy[0];
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t c = 0;
        if (per_channel) {
            const uint32_t row = fdiv(i, elem_per_channel);
            c = row - fdiv(row, num_channel) * num_channel.d;
        }
        float dq;
        const float v = x[i];
        const float d = rl_elem(v, scale[c], offset[c], per_channel, qmin, qmax, rounding, &dq);
        float grad = (float)((v > dq) ? 1 : -1) * g;
        if (d < 0.f) grad = 0.f;
        dx[i] = grad / root;
    }
}

static int validate(int64_t n, int64_t C, int64_t epc, const char* what) {
    if (n <= 0) { set_error("%s: tensor is empty", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (n > 0x7fffffffLL) { set_error("%s: too many elements", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (C > 0 && (epc <= 0 || n % (C * epc) != 0)) {
        set_error("%s: bad channel geometry", what); return PPQHIP_ERR_INVALID_VALUE;
    }
    return PPQHIP_OK;
}

}  // namespace ppqhip

using namespace ppqhip;

extern "C" {

int ppqhip_tensor_clip_t(const float* value, const float* reference, const float* limit, float* out, int64_t n,
                         void* stream) {
    if (int st = validate(n, 0, 1, "tensor_clip_t")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_TENSOR_CLIP, 12.0 * (double)n, s);
    hipLaunchKernelGGL(tensor_clip_kernel, dim3(stream_grid(n, kBlock * 4)), dim3(kBlock), 0, s, value, reference,
                       limit, out, (uint32_t)n, make_fastdiv(1), make_fastdiv(1), 0);
    return finish_launch("tensor_clip_t");
}

int ppqhip_tensor_clip_c(const float* value, const float* reference, const float* limit, float* out, int64_t n,
                         int64_t num_channel, int64_t elem_per_channel, void* stream) {
    if (int st = validate(n, num_channel, elem_per_channel, "tensor_clip_c")) return st;
    if (num_channel <= 0) { set_error("tensor_clip_c: no channels"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_TENSOR_CLIP, 12.0 * (double)n, s);
    hipLaunchKernelGGL(tensor_clip_kernel, dim3(stream_grid(n, kBlock * 4)), dim3(kBlock), 0, s, value, reference,
                       limit, out, (uint32_t)n, make_fastdiv((uint32_t)elem_per_channel),
                       make_fastdiv((uint32_t)num_channel), 1);
    return finish_launch("tensor_clip_c");
}

int ppqhip_rounding_loss(const float* x, const float* scale, const float* offset, float* out, int64_t n,
                         int64_t num_channel, int64_t elem_per_channel, int clip_min, int clip_max, int rounding,
                         void* stream) {
    if (int st = validate(n, num_channel, elem_per_channel, "rounding_loss")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_ROUNDING_LOSS, 4.0 * (double)n, s);
    if (int st = check_hip(hipMemsetAsync(out, 0, sizeof(float), s), "memset loss")) return st;
    const int pc = num_channel > 0;
    hipLaunchKernelGGL(rounding_loss_kernel, dim3(stream_grid(n, kBlock * 8, num_cu() * 4)), dim3(kBlock), 0, s, x, scale,
                       offset, out, (uint32_t)n, make_fastdiv(pc ? (uint32_t)elem_per_channel : 1u),
                       make_fastdiv(pc ? (uint32_t)num_channel : 1u), pc, clip_min, clip_max, rounding,
                       1.0f / sqrtf((float)n));
    return finish_launch("rounding_loss");
}

int ppqhip_rounding_loss_bwd(const float* x, const float* dy, const float* scale, const float* offset, float* dx,
                             int64_t n, int64_t num_channel, int64_t elem_per_channel, int clip_min, int clip_max,
                             int rounding, void* stream) {
    if (int st = validate(n, num_channel, elem_per_channel, "rounding_loss_bwd")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_ROUNDING_LOSS, 8.0 * (double)n, s);
    const int pc = num_channel > 0;
    hipLaunchKernelGGL(rounding_loss_bwd_kernel, dim3(stream_grid(n, kBlock * 4)), dim3(kBlock), 0, s, x, dy, scale,
                       offset, dx, (uint32_t)n, make_fastdiv(pc ? (uint32_t)elem_per_channel : 1u),
                       make_fastdiv(pc ? (uint32_t)num_channel : 1u), pc, clip_min, clip_max, rounding,
                       sqrtf((float)n));
    return finish_launch("rounding_loss_bwd");
}

}  // extern "C"
