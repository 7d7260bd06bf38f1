"""How far are the KL losses of kl_losses_kernel from the reference's arithmetic (oracle.kl_search, pinned to the reference's
hist_to_scale_offset) on histograms with more than 2^24 counts, and how close do neighbouring candidates come?
    python tools/kl_accuracy.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ppq_oracle as O  # noqa: E402
from ppq_amd import CUDA  # noqa: E402

dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
worst_rel, flips, closest = 0.0, 0, 1.0
for trial in range(40):
    n = int(2 ** np.random.default_rng(trial).uniform(20, 25.6))
    z = torch.randn(n, device=dev, generator=g)
    kind = trial % 4
    x = [z, torch.relu(z), torch.relu(z) * torch.rand(n, device=dev, generator=g), z * (torch.rand(n, device=dev, generator=g) < 0.1)][kind]
    hs = float(x.abs().max()) / 2048
    hist = torch.zeros(2048, dtype=torch.int32, device=dev)
    CUDA.Histogram_T(x, hist, hs, True)
    ours = CUDA.KLLosses(hist.reshape(1, -1), 8).cpu().numpy().reshape(-1)
    _, _, losses, best = O.kl_search(hist.cpu().numpy(), hs, return_losses=True)
    ref = np.array([d['kl'] for d in losses])
    rel = np.abs(ours - ref) / np.maximum(np.abs(ref), 1e-300)
    order = np.sort(ref)
    gap = (order[1] - order[0]) / max(abs(order[0]), 1e-300)
    worst_rel, closest = max(worst_rel, float(rel.max())), min(closest, float(gap))
    if int(np.argmin(ours)) != int(np.argmin(ref)): flips += 1
    print(f'trial {trial:2d} kind {kind} n {n:9d}: max rel loss diff {rel.max():.2e}, gap best-vs-second {gap:.2e}, argmin ours {int(np.argmin(ours))} ref {int(np.argmin(ref))}')
print(f'worst relative loss difference {worst_rel:.2e}; smallest best-vs-second gap {closest:.2e}; arg-min flips {flips} / 40')
