/* TEST INFRASTRUCTURE -- host stand-in for <cuda_runtime.h> (see cuda.h beside it). */
#pragma once
#include "cuda.h"
