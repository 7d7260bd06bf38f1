"""Kernel micro-benchmark: GB/s of every hot kernel on A=[1,3,224,224], B=[1,512,56,56], Bx32.
Timing: torch.cuda.Event pairs on the current stream (the stream the library launches on),
`iters` back-to-back launches per measurement -> average launch-to-launch time."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd import CUDA  # noqa: E402


def timeit(fn, iters=50, warmup=5):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--relu', action='store_true')
    ap.add_argument('--dist', default='randn', choices=['randn', 'uniform', 'lanes', 'outliers'],
                    help='input distribution: lanes = every lane of a wave hits its own bin (no LDS conflicts)')
    ap.add_argument('--bins', type=int, default=2048)
    ap.add_argument('--only', type=str, default='')
    ap.add_argument('--rotate', type=int, default=1, help='cycle through this many distinct input tensors (defeats the 256 MiB Infinity Cache)')
    ap.add_argument('--tensors', type=str, default='A,B,Bx8,Bx32')
    args = ap.parse_args()
    dev = 'cuda'
    torch.manual_seed(0)
    shapes = {'A': (1, 3, 224, 224), 'B': (1, 512, 56, 56), 'Bx8': (8, 512, 56, 56), 'Bx32': (32, 512, 56, 56)}
    rows = []
    for name, shp in shapes.items():
        if name not in args.tensors.split(','): continue
        xs = [torch.randn(*shp, device=dev) for _ in range(args.rotate)]
        if args.dist == 'uniform': xs = [torch.rand(*shp, device=dev) * 2 - 1 for _ in range(args.rotate)]
        if args.dist == 'outliers': xs = [t * (1 + 50 * (torch.rand_like(t) < 1e-5)) for t in xs]
        if args.dist == 'lanes':
            nn = xs[0].numel()
            base = ((torch.arange(nn, device=dev) // 4) % 64).float() + 0.5
            xs = [(base / 64.0).reshape(shp) * (1 - 1e-3 * k) for k in range(args.rotate)]
            xs[0].view(-1)[0] = float(args.bins) / 64.0   # abs max -> hist_scale = 1/64, bin = lane
        if args.relu: xs = [torch.relu(t) for t in xs]
        x = xs[0]
        counter = [0]

        def X():
            counter[0] += 1
            return xs[counter[0] % len(xs)]
        n = x.numel()
        C = shp[1]
        s1 = torch.tensor([0.03], device=dev); o1 = torch.zeros(1, device=dev)
        sc = torch.rand(C, device=dev) * 0.05 + 0.01; oc = torch.randint(0, 255, [C], device=dev).float()
        hist = torch.zeros(args.bins, dtype=torch.int32, device=dev)
        mm = torch.tensor([float('inf'), float('-inf')], device=dev)
        mins = torch.full([C], float('inf'), device=dev); maxs = torch.full([C], float('-inf'), device=dev)
        hs = float(x.abs().max()) / args.bins
        dyv = torch.rand_like(x)
        lo, hi = float(x.min()), float(x.max())
        cases = {
            'fq_linear_t': (8, lambda: CUDA.LinearQuantize_T(X(), s1, o1, -128, 127, 0)),
            'fq_linear_c': (8, lambda: CUDA.LinearQuantize_C(X(), sc, oc, 1, 0, 255, 0)),
            'fq_float_t': (8, lambda: CUDA.FloatingQuantize_T(X(), s1, o1)),
            'hist_sym_t': (4, lambda: CUDA.Histogram_T(X(), hist, hs)),
            'hist_asym_t': (4, lambda: CUDA.Histogram_Asymmetric_T(lo, hi, X(), hist)),
            'minmax_t': (4, lambda: CUDA.MinMax_T(X(), mm)),
            'minmax_c': (4, lambda: CUDA.MinMax_C(X(), 1, mins, maxs)),
            'quantile_t': (4, lambda: CUDA.Quantile(X(), 0.9999)),
            'lsq_bwd_t': (12, lambda: CUDA.LinearQuantize_T_B(X(), s1, o1, dyv, -128, 127, 0)),
            'lsq_bwd_c': (12, lambda: CUDA.LinearQuantize_C_B(X(), sc, oc, dyv, 0, 255, 1, 0)),
            'torch copy (ref)': (8, lambda: X().clone()),
            'torch abs().max (ref)': (4, lambda: X().abs().max()),
        }
        for k, (bpe, fn) in cases.items():
            if args.only and not any(o in k for o in args.only.split(',')): continue
            t = timeit(fn, iters=200 if n < 10_000_000 else 30)
            rows.append({'kernel': k, 'tensor': name, 'us': round(t * 1e6, 2), 'GBps': round(bpe * n / t / 1e9, 1)})
            print(f'{k:24s} {name:5s} {t*1e6:10.2f} us  {bpe*n/t/1e9:9.1f} GB/s', flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    tag = ('relu' if args.relu else 'randn') + f'_{args.bins}'
    json.dump(rows, open(f'gpurun_out/microbench_{tag}.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
