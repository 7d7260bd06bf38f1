"""Soak of the round-4 multi-tensor entry points on a GPU: random job counts, shapes, channel axes and (un)aligned storage;
ppqhip_fq_linear_c_bwd_multi vs ppqhip_fq_linear_c_bwd per tensor (grad_x bitwise, grad_s to summation tolerance) and
ppqhip_minmax_c_multi vs ppqhip_minmax_c (bitwise), fresh and accumulating.   python tools/multi_soak.py [rounds] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd import CUDA  # noqa: E402

dev = 'cuda'
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def tensor(shape, unaligned):
    n = int(np.prod(shape))
    base = torch.randn(n + 3, device=dev) * float(rng.uniform(0.05, 3))
    return (base[1:n + 1] if unaligned else base[:n]).reshape(shape)      # [1:] = 4-B aligned only


for r in range(rounds):
    jobs = int(rng.integers(1, 41))
    xs, dys, ss, os_, axes = [], [], [], [], []
    for _ in range(jobs):
        nd = int(rng.integers(1, 5))
        shape = tuple(int(v) for v in rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 27, 64, 65, 144, 256, 577], size=nd))
        if int(np.prod(shape)) > 2_000_000: shape = shape[:2]
        axis = int(rng.integers(0, len(shape)))
        un = bool(rng.integers(0, 4) == 0)
        xs.append(tensor(shape, un)); dys.append(tensor(shape, bool(rng.integers(0, 4) == 0)))
        C = shape[axis]
        ss.append(torch.rand(C, device=dev) * 0.1 + 0.01); os_.append(torch.randint(-3, 4, [C], device=dev).float())
        axes.append(axis)
    qmin, qmax = [(-8, 7), (-128, 127), (0, 255)][r % 3]
    rounding = int(rng.integers(0, 8))
    gxs, gss = CUDA.LinearQuantize_C_B_Multi(xs, ss, os_, dys, [qmin] * jobs, [qmax] * jobs, axes, rounding)
    for x, dy, s, o, a, gx, gs in zip(xs, dys, ss, os_, axes, gxs, gss):
        wx, ws = CUDA.LinearQuantize_C_B(x, s, o, dy, qmin, qmax, a, rounding)
        tol = 1e-5 * float(dy.abs().sum()) / max(1, x.shape[a]) / np.sqrt(x.numel() * max(qmax, 1)) * 16 + 1e-7
        if not torch.equal(gx, wx) or not torch.allclose(gs, ws, rtol=5e-4, atol=tol):
            bad += 1; print('LSQ mismatch', r, tuple(x.shape), a, float((gs - ws).abs().max()), tol)
    fresh = [CUDA.minmax_c_fresh_ok(x, a) and bool(rng.integers(0, 2)) for x, a in zip(xs, axes)]
    mins = [torch.full([x.shape[a]], 9.0 if f else float('inf'), device=dev) for x, a, f in zip(xs, axes, fresh)]
    maxs = [torch.full([x.shape[a]], -9.0 if f else float('-inf'), device=dev) for x, a, f in zip(xs, axes, fresh)]
    CUDA.MinMax_C_Multi(xs, axes, mins, maxs, fresh)
    for x, a, mn, mx in zip(xs, axes, mins, maxs):
        wmn = torch.full([x.shape[a]], float('inf'), device=dev); wmx = torch.full([x.shape[a]], float('-inf'), device=dev)
        CUDA.MinMax_C(x, a, wmn, wmx)
        if not (torch.equal(mn, wmn) and torch.equal(mx, wmx)):
            bad += 1; print('minmax mismatch', r, tuple(x.shape), a)
torch.cuda.synchronize()
print(f'rounds {rounds} mismatches {bad}')
