"""Single-tensor histogram launches on B x m under the developer knobs of a -DPPQHIP_DEV_KNOBS build (PPQHIP_LIBRARY=variants/lib_dev.so):
rows form through the persistent kernel / the ping-pong form of hist_small_kernel (K = 2, 4) / the all-loads-up-front form, and the
one-shot form.  Run under rocprofv3 --kernel-trace; tools/kernel_times.py gives the medians per (kernel, grid).
    python tools/hist_stream_bench.py <m> [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd._lib import lib  # noqa: E402
from ppq_amd import CUDA  # noqa: E402

dev = torch.device('cuda')
m = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = m * 512 * 56 * 56
bins = 2048
g = torch.Generator(device=dev).manual_seed(0)
rot = max(2, min(8, (1 << 30) // (4 * n)))
xs = [torch.randn(n, device=dev, generator=g) for _ in range(rot)]
hs = float(xs[0].abs().max()) / bins
rows = torch.zeros(CUDA.hist_rows(), bins, dtype=torch.int32, device=dev)
hist = torch.zeros(bins, dtype=torch.int32, device=dev)
ws = torch.empty(int(lib.ppqhip_hist_workspace_bytes(n, bins)) + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
KNOBS = ('PPQHIP_DEV_HIST_STREAM', 'PPQHIP_DEV_HIST_SMALL_ELEMS', 'PPQHIP_DEV_HIST_WG', 'PPQHIP_DEV_HIST_ONESHOT')
cases = [('rows persistent', {'PPQHIP_DEV_HIST_STREAM': '0'}, True),
         ('rows stream K=2', {'PPQHIP_DEV_HIST_STREAM': '2'}, True),
         ('rows stream K=4', {'PPQHIP_DEV_HIST_STREAM': '4'}, True),
         ('rows all-up-front K<=8', {'PPQHIP_DEV_HIST_STREAM': '0', 'PPQHIP_DEV_HIST_SMALL_ELEMS': str(0x7fffffff)}, True),
         ('oneshot default', {}, False),
         ('oneshot rows8', {'PPQHIP_DEV_HIST_ONESHOT': '8'}, False),
         ('oneshot rows2', {'PPQHIP_DEV_HIST_ONESHOT': '2'}, False),
         ('oneshot direct atomics', {'PPQHIP_DEV_HIST_ONESHOT': '-1'}, False),
         ('oneshot direct, grid 256', {'PPQHIP_DEV_HIST_ONESHOT': '-1', 'PPQHIP_DEV_HIST_WG': '256'}, False),
         ('oneshot reduce launch (r5)', {'PPQHIP_DEV_HIST_ONESHOT': '0'}, False)]
for name, env, is_rows in cases:
    for k in KNOBS: os.environ.pop(k, None)
    os.environ.update(env)
    rows.zero_(); hist.zero_()
    for i in range(iters):
        x = xs[i % rot]
        if is_rows: rc = lib.ppqhip_hist_sym_t_rows(x.data_ptr(), n, hs, 1, rows.data_ptr(), bins, st)
        else: rc = lib.ppqhip_hist_sym_t(x.data_ptr(), n, hs, 1, hist.data_ptr(), bins, ws.data_ptr(), st)
        assert rc == 0
    torch.cuda.synchronize()
    total = int(rows.sum().item()) if is_rows else int(hist.sum().item())
    # a checksum per case: every variant must have counted the same elements
    print(f'x{m} {name:28s} counted {total}', flush=True)
    lib.ppqhip_hist_sym_t_rows(xs[0].data_ptr(), 4096, hs, 1, rows.data_ptr(), bins, st)     # a separator launch in the trace (grid 1)
