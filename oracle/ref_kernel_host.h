/* TEST INFRASTRUCTURE ONLY -- a sequential host stand-in for the CUDA execution model, just large enough to
 * run the reference's `__global__` kernel BODIES (extracted at build time by extract_kernels.awk from the
 * files where they lie; see oracle/Makefile) under g++.  Written for this repository; contains nothing of
 * the CUDA toolkit or of the reference.
 *
 *   blockIdx / threadIdx / blockDim / gridDim   plain globals, set by launch() below
 *   launch(grid, block, body [, slot])           runs body() once per (block, thread); inside one block the
 *                                               threads run in DESCENDING order, so thread 0 runs last
 *   BlockReduceSum<T>(v)                         the k-th call of a thread adds v to the block's k-th sum and
 *                                               returns the running sum: thread 0 (last) therefore receives
 *                                               the block total, which is all the kernels use it for
 *                                               (`if (threadIdx.x == 0) atomicAdd(...)`).  Accumulated in
 *                                               double; the device's shuffle tree sums in float in another
 *                                               order, so scale gradients are compared with a tolerance,
 *                                               masks / grad_x / per-element terms bit for bit
 *   atomicAdd, __syncthreads, __float2int_rn    the obvious sequential meanings; __float2int_rn saturates and
 *                                               maps NaN to 0 like cvt.rni.s32.f32
 *
 * Host-vs-device caveat (same as ref_common_shim.cc): an `int b = floor(...)` of a value outside int32 is
 * undefined on the host (x86: INT_MIN) and saturating on the device; comparisons keep inputs in range. */
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

struct ref_dim3 { unsigned x = 1, y = 1, z = 1; };
static ref_dim3 blockIdx, threadIdx, blockDim, gridDim;

static std::vector<double> g_block_sums;    /* k-th BlockReduceSum of the running block */
static int g_call = 0;                      /* BlockReduceSum calls made by the running thread */
static float* g_partials = nullptr;         /* optional: what each thread handed to its FIRST BlockReduceSum */
static int64_t g_slot = -1;

template <typename T> static inline T BlockReduceSum(T val) {
    if ((size_t)g_call >= g_block_sums.size()) g_block_sums.resize(g_call + 1, 0.0);
    g_block_sums[g_call] += (double)val;
    if (g_partials && g_slot >= 0 && g_call == 0) g_partials[g_slot] = (float)val;
    return (T)g_block_sums[g_call++];
}

static inline void __syncthreads() {}
static inline int atomicAdd(int* a, int v) { int old = *a; *a += v; return old; }
static inline float atomicAdd(float* a, float v) { float old = *a; *a += v; return old; }

static inline int __float2int_rn(float v) {
    if (v != v) return 0;
    float r = nearbyintf(v);
    if (r >= 2147483648.0f) return std::numeric_limits<int>::max();
    if (r <= -2147483648.0f) return std::numeric_limits<int>::min();
    return (int)r;
}

struct no_slot { int64_t operator()() const { return -1; } };

template <typename Body, typename Slot = no_slot>
static void launch(unsigned gx, unsigned gy, unsigned block, Body body, Slot slot = Slot()) {
    gridDim.x = gx; gridDim.y = gy; gridDim.z = 1;
    blockDim.x = block; blockDim.y = 1; blockDim.z = 1;
    for (unsigned bx = 0; bx < gx; bx++)
        for (unsigned by = 0; by < gy; by++) {
            blockIdx.x = bx; blockIdx.y = by; blockIdx.z = 0;
            g_block_sums.clear();
            for (int t = (int)block - 1; t >= 0; t--) {
                threadIdx.x = (unsigned)t; threadIdx.y = 0; threadIdx.z = 0;
                g_call = 0;
                g_slot = slot();
                body();
            }
        }
}
