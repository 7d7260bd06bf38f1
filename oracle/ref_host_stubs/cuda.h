/* TEST INFRASTRUCTURE -- host stand-in for <cuda.h>, so that the reference's header-only device code
 * (ppq/csrc/cuda/common.cuh) compiles with g++ exactly where it lies (oracle/Makefile target `ref`).
 * Written for this repository; contains nothing of the CUDA toolkit or of the reference. */
#pragma once
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
template <typename T> static inline T __ldg(const T* p) { return *p; }
