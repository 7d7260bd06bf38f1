"""How often the hinted single-tensor quantile is SETTLED from its hint when the activation's scale moves from batch to batch.
The wanted keys sit 3.7 sigma out (q = 0.9999): the number of keys beyond a fixed threshold moves with the 14th power of the scale, so a
list aimed at 1.5 x the wanted keys is used up by a 3 % smaller batch.  Per size and jitter (scale_b = exp(N(0, jitter))): calls settled
from the hint out of 200, and the event-timed mean per call.    python tools/quantile_drift.py [sizes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd import CUDA  # noqa: E402
from ppq_amd.ffi import quantile_hint  # noqa: E402

dev = torch.device('cuda')
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '1,8').split(',')]
g = torch.Generator(device=dev).manual_seed(0)
rng = np.random.default_rng(0)
calls = 200
for m in sizes:
    n = m * 512 * 56 * 56
    pool = 8 if m <= 8 else 4
    base = [torch.randn(n, device=dev, generator=g) for _ in range(pool)]
    for kind in ('randn', 'relu'):
        for jitter in (0.0, 0.02, 0.05, 0.1, 0.2):
            scales = np.exp(rng.normal(0.0, jitter, pool))
            xs = [(torch.relu(b) if kind == 'relu' else b) * float(s) for b, s in zip(base, scales)]
            order = rng.integers(0, pool, calls)
            hint = quantile_hint(dev)
            CUDA.Quantile_Hinted(xs[0], 0.9999, hint)                         # the first call: the general sequence
            torch.cuda.synchronize()
            used0 = int(hint.cpu()[7])
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in order: CUDA.Quantile_Hinted(xs[int(i)], 0.9999, hint)
            e1.record(); torch.cuda.synchronize()
            print(f'x{m} {kind:5s} jitter {jitter:4.2f}: settled from the hint {int(hint.cpu()[7]) - used0:3d} / {calls}, {e0.elapsed_time(e1) * 1e3 / calls:6.1f} us per call', flush=True)
            del xs

# the multi-tensor sequence (what ppq_amd's own percentile observers use: one sequence per forward, hints per observer)
if len(sys.argv) > 2 and sys.argv[2] == 'multi':
    jobs, m, pool = 16, 4, 4
    n = m * 512 * 56 * 56
    base = [[torch.relu(torch.randn(n, device=dev, generator=g)) for _ in range(pool)] for _ in range(jobs)]
    for jitter in (0.0, 0.02, 0.05, 0.1, 0.2):
        xs = [[b * float(np.exp(rng.normal(0.0, jitter))) for b in bs] for bs in base]
        hints = [quantile_hint(dev) for _ in range(jobs)]
        CUDA.Quantile_Multi([x[0] for x in xs], 0.9999, None, hints)
        torch.cuda.synchronize()
        used0 = sum(int(h.cpu()[7]) for h in hints)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        picks = rng.integers(0, pool, (50, jobs))
        e0.record()
        for row in picks: CUDA.Quantile_Multi([xs[j][int(row[j])] for j in range(jobs)], 0.9999, None, hints)
        e1.record(); torch.cuda.synchronize()
        used = sum(int(h.cpu()[7]) for h in hints) - used0
        print(f'multi: {jobs} x (x{m}, relu) jitter {jitter:4.2f}: job-calls settled from their hint {used:3d} / {50 * jobs}, {e0.elapsed_time(e1) * 1e3 / 50:7.1f} us per sequence '
              f'({jobs * n * 4 / (e0.elapsed_time(e1) * 1e-3 / 50) / 1e12:.2f} TB/s)', flush=True)
        del xs
