// floor_kernels.hip -- MEASUREMENT AID, not part of libppq_hip.so: the cheapest kernels that do what each phase of the
// single-tensor launches on B = [1,512,56,56] must do, so that `rocprofv3 --kernel-trace` prices the library's kernels against
// a floor measured under the same tracer on the same box ("Next round" item 1).  Built by tools/floor/build.sh into
// tools/floor/libfloor.so; driven by tools/floor_table.py.
//
//   floor_empty    : nothing (dispatch + wave launch + end of kernel)
//   floor_read     : every workgroup loads its contiguous share with 16-B loads, all in flight, folds a max, ONE 4-B store per WG
//   floor_copy     : out = x (16-B loads / stores), one tile per workgroup; store flavours plain / nontemporal
//   floor_atomic   : G workgroups x `bins` counters: one device atomic per non-skipped counter into (a) ONE shared row,
//                    (b) a row per blockIdx % 8, (c) a row per workgroup -- the cross-workgroup combine of a histogram, alone
//   floor_read_atomic : floor_read followed by floor_atomic's flush in the same kernel (a histogram without the binning)
//   floor_ticket   : one RETURNING device atomic per workgroup; the last arriver stores one word (the "last block folds" chain)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void floor_empty_kernel() {}
__global__ void floor_marker_kernel() {}            // grid size = case id: splits the kernel trace into cases

template <int BLOCK, int U, bool NT>
__global__ __launch_bounds__(BLOCK) void floor_read_kernel(const float4* __restrict__ x, uint32_t nvec, float* __restrict__ sink) {
    __shared__ float part[BLOCK / 64];
    // contiguous share per workgroup, walked in tiles of BLOCK * U float4 (the library's traversal)
    const uint32_t tile = BLOCK * U;
    const uint32_t tiles = (nvec + tile - 1) / tile;
    const uint32_t q = tiles / gridDim.x, rem = tiles - q * gridDim.x;        // balanced split, one 32-bit division
    const uint32_t t0 = blockIdx.x * q + min(blockIdx.x, rem), t1 = t0 + q + (blockIdx.x < rem ? 1u : 0u);
    float m = 0.f;
    for (uint32_t t = t0; t < t1; t++) {
        float4 a[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const uint32_t v = min(t * tile + k * BLOCK + threadIdx.x, nvec - 1);
            if (NT) { const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(x + v)); a[k] = make_float4(q.x, q.y, q.z, q.w); }
            else a[k] = x[v];
        }
#pragma unroll
        for (int k = 0; k < U; k++) m = fmaxf(m, fmaxf(fmaxf(fabsf(a[k].x), fabsf(a[k].y)), fmaxf(fabsf(a[k].z), fabsf(a[k].w))));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BLOCK / 64; w++) m = fmaxf(m, part[w]);
        sink[blockIdx.x] = m;
    }
}

template <int BLOCK, int U, int STORE>     // STORE: 0 plain, 1 nontemporal
__global__ __launch_bounds__(BLOCK) void floor_copy_kernel(const float4* __restrict__ x, float4* __restrict__ out, uint32_t nvec) {
    const uint32_t base = blockIdx.x * (BLOCK * U) + threadIdx.x;
    float4 a[U];
#pragma unroll
    for (int k = 0; k < U; k++) a[k] = x[min(base + k * BLOCK, nvec - 1)];
#pragma unroll
    for (int k = 0; k < U; k++) {
        const uint32_t v = base + k * BLOCK;
        if (v < nvec) {
            if (STORE == 1) { v4f q = {a[k].x, a[k].y, a[k].z, a[k].w}; __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(out + v)); }
            else out[v] = a[k];
        }
    }
}

// rows_mode 0: dst[bins]; 1: dst[8][bins] by blockIdx % 8; 2: dst[grid][bins].  keep: a counter is flushed when (b * 2654435761u >> 16) % 16 < keep
__device__ __forceinline__ void flush_atomics(int* __restrict__ dst, int bins, int rows_mode, int keep, int block) {
    int* row = dst + (rows_mode == 0 ? 0 : (rows_mode == 1 ? (blockIdx.x & 7) : blockIdx.x)) * (size_t)bins;
    for (int b = threadIdx.x; b < bins; b += block) {
        const unsigned h = ((unsigned)(b + 17 * blockIdx.x) * 2654435761u >> 16) & 15u;
        if ((int)h < keep) atomicAdd(&row[b], 1);
    }
}

__global__ void floor_atomic_kernel(int* __restrict__ dst, int bins, int rows_mode, int keep) {
    flush_atomics(dst, bins, rows_mode, keep, blockDim.x);
}

template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void floor_read_atomic_kernel(const float4* __restrict__ x, uint32_t nvec, int* __restrict__ dst, int bins,
                                                                  int rows_mode, int keep) {
    const uint32_t tile = BLOCK * U;
    const uint32_t tiles = (nvec + tile - 1) / tile;
    const uint32_t q = tiles / gridDim.x, rem = tiles - q * gridDim.x;        // balanced split, one 32-bit division
    const uint32_t t0 = blockIdx.x * q + min(blockIdx.x, rem), t1 = t0 + q + (blockIdx.x < rem ? 1u : 0u);
    float m = 0.f;
    for (uint32_t t = t0; t < t1; t++) {
        float4 a[U];
#pragma unroll
        for (int k = 0; k < U; k++) a[k] = x[min(t * tile + k * BLOCK + threadIdx.x, nvec - 1)];
#pragma unroll
        for (int k = 0; k < U; k++) m = fmaxf(m, fmaxf(fmaxf(fabsf(a[k].x), fabsf(a[k].y)), fmaxf(fabsf(a[k].z), fabsf(a[k].w))));
    }
    if (m > 1e30f) keep = 16;                 // data dependent: the loads cannot be dropped, the flush waits for them
    __syncthreads();
    flush_atomics(dst, bins, rows_mode, keep, BLOCK);
}

__global__ void floor_ticket_kernel(unsigned* __restrict__ counter, unsigned* __restrict__ out) {
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(counter, 1u);
        if (t == gridDim.x - 1) { out[0] = t; atomicExch(counter, 0u); }
    }
}

extern "C" {

int floor_marker(int case_id, void* stream) {
    hipLaunchKernelGGL(floor_marker_kernel, dim3(case_id), dim3(64), 0, (hipStream_t)stream);
    return (int)hipGetLastError();
}

int floor_empty(int grid, int block, void* stream) {
    hipLaunchKernelGGL(floor_empty_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream);
    return (int)hipGetLastError();
}

#define READ_CASE(B, U)                                                                                                  \
    if (block == B && unroll == U) {                                                                                     \
        if (nt) hipLaunchKernelGGL((floor_read_kernel<B, U, true>), dim3(grid), dim3(B), 0, s, (const float4*)x, nvec, sink); \
        else hipLaunchKernelGGL((floor_read_kernel<B, U, false>), dim3(grid), dim3(B), 0, s, (const float4*)x, nvec, sink); \
        return (int)hipGetLastError();                                                                                   \
    }
// grid <= 0: one tile per workgroup
int floor_read(const float* x, int64_t n, float* sink, int grid, int block, int unroll, int nt, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nvec = (uint32_t)(n / 4);
    if (grid <= 0) grid = (int)((nvec + block * unroll - 1) / (block * unroll));
    READ_CASE(256, 1) READ_CASE(256, 2) READ_CASE(256, 4) READ_CASE(512, 1) READ_CASE(512, 2) READ_CASE(512, 4) READ_CASE(1024, 1) READ_CASE(1024, 2)
    return -1;
}

#define COPY_CASE(B, U)                                                                                                  \
    if (block == B && unroll == U) {                                                                                     \
        if (store == 1) hipLaunchKernelGGL((floor_copy_kernel<B, U, 1>), dim3(grid), dim3(B), 0, s, (const float4*)x, (float4*)out, nvec); \
        else hipLaunchKernelGGL((floor_copy_kernel<B, U, 0>), dim3(grid), dim3(B), 0, s, (const float4*)x, (float4*)out, nvec); \
        return (int)hipGetLastError();                                                                                   \
    }
int floor_copy(const float* x, float* out, int64_t n, int block, int unroll, int store, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nvec = (uint32_t)(n / 4);
    const int grid = (int)((nvec + block * unroll - 1) / (block * unroll));
    COPY_CASE(256, 1) COPY_CASE(256, 2) COPY_CASE(256, 4) COPY_CASE(512, 1) COPY_CASE(512, 2) COPY_CASE(1024, 1)
    return -1;
}

int floor_atomic(int* dst, int bins, int grid, int block, int rows_mode, int keep, void* stream) {
    hipLaunchKernelGGL(floor_atomic_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, dst, bins, rows_mode, keep);
    return (int)hipGetLastError();
}

int floor_read_atomic(const float* x, int64_t n, int* dst, int bins, int grid, int rows_mode, int keep, void* stream) {
    hipLaunchKernelGGL((floor_read_atomic_kernel<512, 2>), dim3(grid), dim3(512), 0, (hipStream_t)stream, (const float4*)x, (uint32_t)(n / 4), dst,
                       bins, rows_mode, keep);
    return (int)hipGetLastError();
}

int floor_ticket(unsigned* counter, unsigned* out, int grid, int block, void* stream) {
    hipLaunchKernelGGL(floor_ticket_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, counter, out);
    return (int)hipGetLastError();
}

}  // extern "C"
