"""Wall time per launch of the persistent-accumulator observers (rows / slots mode) over tensor sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppq_amd import CUDA
dev = 'cuda'
for m in (1, 2, 4, 8, 16, 32):          # x 1.6M elements (6.4 MB)
    xs = [torch.relu(torch.randn(m, 512, 56, 56, device=dev)) for _ in range(max(2, 64 // m))]
    rows = torch.zeros(CUDA.hist_rows(), 2048, dtype=torch.int32, device=dev)
    slots = torch.empty(CUDA.minmax_slots(), 2, device=dev); slots[:, 0] = float('inf'); slots[:, 1] = float('-inf')
    hs = float(xs[0].abs().max()) / 2048
    for name, fn in (('hist_rows', lambda x: CUDA.Histogram_T_Rows(x, rows, hs)), ('minmax_slots', lambda x: CUDA.MinMax_T_Slots(x, slots))):
        for i in range(20): fn(xs[i % len(xs)])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N = 400
        for i in range(N): fn(xs[i % len(xs)])
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / N * 1e6
        print(f'{name:14s} {m * 6.4:7.1f} MB  {us:8.2f} us  {m * 6.4e6 / us / 1e6:8.1f} GB/s')
