/* TEST INFRASTRUCTURE -- host stand-in for <ATen/cuda/CUDAContext.h> (nothing of it is needed on the host). */
#pragma once
