"""Host-side cost of one facade call (no sync): how much Python sits in front of a launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppq_amd import CUDA, _lib
from ppq_amd.ffi import _stream
x = torch.randn(4096, device='cuda'); s = torch.tensor([0.1], device='cuda'); o = torch.zeros(1, device='cuda')
mm = torch.tensor([float('inf'), float('-inf')], device='cuda')
hist = torch.zeros(2048, dtype=torch.int32, device='cuda')
def t(fn, n=20000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0; torch.cuda.synchronize()
    return dt / n * 1e6
print('LinearQuantize_T     %.2f us' % t(lambda: CUDA.LinearQuantize_T(x, s, o, -128, 127, 0)))
print('MinMax_T             %.2f us' % t(lambda: CUDA.MinMax_T(x, mm)))
print('Histogram_T          %.2f us' % t(lambda: CUDA.Histogram_T(x, hist, 0.01)))
print('torch.empty_like     %.2f us' % t(lambda: torch.empty_like(x)))
print('x.contiguous()       %.2f us' % t(lambda: x.contiguous()))
print('_stream()            %.2f us' % t(_stream))
print('raw stream           %.2f us' % t(lambda: torch._C._cuda_getCurrentRawStream(0)))
lib = _lib.lib
xp, sp, op_ = x.data_ptr(), s.data_ptr(), o.data_ptr(); out = torch.empty_like(x); outp = out.data_ptr()
print('ctypes call only     %.2f us' % t(lambda: lib.ppqhip_fq_linear_t(xp, sp, op_, outp, 4096, -128, 127, 0, 0)))
print('torch relu (ref)     %.2f us' % t(lambda: torch.relu(x)))
print('x.data_ptr()         %.2f us' % t(lambda: x.data_ptr()))
