"""Device time of the rows-mode histogram kernel over a range of tensor sizes (run under rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppq_amd import CUDA
for m in (1, 2, 4, 8, 16, 32):          # x 1.6M elements
    x = torch.relu(torch.randn(m, 512, 56, 56, device='cuda'))
    rows = torch.zeros(CUDA.hist_rows(), 2048, dtype=torch.int32, device='cuda')
    hs = float(x.abs().max()) / 2048
    for _ in range(50): CUDA.Histogram_T_Rows(x, rows, hs)
    torch.cuda.synchronize()
