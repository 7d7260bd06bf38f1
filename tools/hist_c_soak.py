"""Soak test of the per-channel histogram kernels (hist.hip: hist_c_channel_kernel / _scalar_kernel / hist_c_global_kernel):
random shapes, channel axes, bin counts, symmetric (shared and per-channel scales) and asymmetric (per-channel ranges) rules,
clipping on and off -- every channel's row must equal the per-tensor kernel run on that channel's slice (itself pinned to the
oracle bit for bit by the test-suite), including accumulation into a non-zero histogram.
    python tools/hist_c_soak.py [rounds] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd import CUDA  # noqa: E402

dev = 'cuda'
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
g = torch.Generator(device=dev).manual_seed(seed)
bad = 0
for r in range(rounds):
    nd = int(rng.integers(2, 5))
    shape = [int(rng.choice([1, 2, 3, 4, 7, 8, 14, 16, 28, 33, 56, 64, 100])) for _ in range(nd)]
    if rng.random() < 0.3: shape[-1] = int(rng.choice([49, 196, 784, 3136, 4099]))
    axis = int(rng.integers(0, nd))
    while int(np.prod(shape)) > 8_000_000: shape[int(rng.integers(0, nd))] = 2
    C = shape[axis]
    bins = int(rng.choice([128, 512, 2048, 4096]))
    x = torch.randn(shape, device=dev, generator=g) * float(rng.uniform(0.5, 4))
    if rng.random() < 0.3: x = torch.relu(x)
    clip = bool(rng.random() < 0.5)
    mode = int(rng.integers(0, 3))
    hist = torch.randint(0, 5, [C, bins], device=dev, dtype=torch.int32, generator=g)
    want = hist.clone()
    xs = x.movedim(axis, 0).reshape(C, -1).contiguous()
    if mode == 0:
        hs = float(x.abs().max()) / bins * float(rng.uniform(0.6, 1.2)) + 1e-6
        CUDA.Histogram_C(x, axis, hist, hs, clip)
        for c in range(C): CUDA.Histogram_T(xs[c], want[c], hs, clip)
    elif mode == 1:
        scales = (xs.abs().amax(dim=1) / bins * float(rng.uniform(0.6, 1.2)) + 1e-6).contiguous()
        CUDA.Histogram_C_Scales(x, axis, hist, scales, clip)
        for c in range(C): CUDA.Histogram_T(xs[c], want[c], float(scales[c]), clip)
    else:
        mins = (xs.amin(dim=1) * float(rng.uniform(0.7, 1.1)) - 1e-3).contiguous(); maxs = (xs.amax(dim=1) * float(rng.uniform(0.7, 1.1)) + 1e-3).contiguous()
        CUDA.Histogram_Asymmetric_C_Ranges(x, axis, hist, mins, maxs, clip)
        for c in range(C): CUDA.Histogram_Asymmetric_T(float(mins[c]), float(maxs[c]), xs[c], want[c], clip)
    if not torch.equal(hist, want):
        bad += 1
        d = (hist != want).nonzero()
        print('MISMATCH', r, shape, axis, bins, mode, clip, 'first', d[0].tolist(), int(hist[tuple(d[0])]), int(want[tuple(d[0])]), flush=True)
print(f'rounds {rounds} mismatches {bad} (seed {seed})')
sys.exit(1 if bad else 0)
