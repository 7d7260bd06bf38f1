"""The single hinted quantile (quantile.hip, two launches) on B x {multiples}: `iters` calls per size over rotating randn / ReLU
inputs with one hint; prints how many calls the hint settled.  Run under rocprofv3 --kernel-trace for per-kernel durations
(tools/kernel_times.py).    python tools/quantile_hot_bench.py [sizes] [iters] [relu | cold]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd import CUDA  # noqa: E402
from ppq_amd.ffi import quantile_hint  # noqa: E402

dev = torch.device('cuda')
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '1,32').split(',')]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
relu = len(sys.argv) > 3 and sys.argv[3] == 'relu'
cold = len(sys.argv) > 3 and sys.argv[3] == 'cold'       # the hint is zeroed before every call: the select launch's exact passes
g = torch.Generator(device=dev).manual_seed(0)
for m in sizes:
    n = m * 512 * 56 * 56
    rot = max(2, min(8, (1 << 30) // (4 * n)))
    xs = [torch.randn(n, device=dev, generator=g) for _ in range(rot)]
    if relu: xs = [torch.relu(x) for x in xs]
    hint = quantile_hint(dev)
    for i in range(iters):
        if cold: hint.zero_()
        CUDA.Quantile_Hinted(xs[i % rot], 0.9999, hint)
    torch.cuda.synchronize()
    print(f'x{m}: n={n} calls={iters} settled_from_hint={int(hint.cpu()[7])} hint={hint.cpu().tolist()}', flush=True)
