"""Soak test of quantile.hip on a GPU: random job mixes, sizes, distributions, q and hint states, every result compared with
a device sort (the reference's index rule, sort.cu:13-19) -- while a second stream keeps the chip unevenly busy, which is
where a missing release / acquire in the exact passes' "last workgroup" tails would show (MI355X_MICROARCH.md: test every
hand-off under uneven load).
    python tools/quantile_soak.py [rounds] [seed] [big]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd import CUDA  # noqa: E402
from ppq_amd.ffi import quantile_hint  # noqa: E402

dev = torch.device('cuda')
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = len(sys.argv) > 3 and sys.argv[3] == 'big'        # also 25 M / 51 M element tensors (the sharded lists, many slices per job)
rng = np.random.default_rng(seed)
g = torch.Generator(device=dev).manual_seed(seed)


def make(n, kind):
    z = torch.randn(n, device=dev, generator=g)
    if kind == 0: return z * float(rng.uniform(0.1, 30)) + float(rng.uniform(-5, 5))
    if kind == 1: return torch.relu(z)
    if kind == 2: return torch.clamp(z * 4, 0, 6)
    if kind == 3: return torch.clamp(z * 4, 0.1, 5.3)
    if kind == 4: return torch.round(z * 40).clamp(-128, 127)
    if kind == 5: return z / (torch.rand(n, device=dev, generator=g) + 1e-3)
    if kind == 6: return torch.full([n], 1.25, device=dev)
    t = z.clone(); t[:: max(1, n // 37)] *= 1e4
    return t


def want(x, q):
    n = x.numel()
    s = torch.sort(x.flatten())[0]
    qf = np.float32(q)
    k_hi = min(max(int(np.rint(np.float32(n) * qf)), 0), n - 1)
    k_lo = min(max(int(np.rint(np.float32(n) * (np.float32(1) - qf))), 0), n - 1)
    return torch.stack([s[k_hi], s[k_lo]])


# uneven background load on another stream: large copies + small kernels, started before every call
side = torch.cuda.Stream()
junk_a = torch.empty(96 << 20, device=dev); junk_b = torch.empty_like(junk_a)


def disturb():
    with torch.cuda.stream(side):
        for _ in range(int(rng.integers(1, 4))):
            junk_b.copy_(junk_a)
            junk_a[: int(rng.integers(1, 1 << 20))].add_(1.0)


streams = {}     # persistent "observers": (kind, n) -> hint, so that some jobs run hot
bad = 0
calls = 0
t0 = time.time()
for r in range(rounds):
    jobs = int(rng.choice([1, 1, 2, 3, 8, 24, 60]))
    q = float(rng.choice([0.9999, 0.9999, 0.999, 0.99, 0.5, 1.0, 0.0]))
    xs, hints, keys = [], [], []
    for _ in range(jobs):
        n = int(rng.choice([1, 7, 300, 5000, 70_001, 262_144, 1_000_003, 3_145_768, 6_422_528] + ([25_690_112, 51_380_224] if big else [])))
        kind = int(rng.integers(0, 8))
        x = make(n, kind)
        if rng.random() < 0.15 and n > 8: x = x[1:]                    # 4-B aligned only
        key = (kind, x.numel(), q)
        mode = rng.random()
        if mode < 0.5: h = streams.setdefault(key, quantile_hint(dev))       # an observer's hint (maybe already valid)
        elif mode < 0.6:                                                     # garbage
            h = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, 8, dtype=np.int64).astype(np.int32)).to(dev)
        else: h = None
        xs.append(x); hints.append(h); keys.append(key)
    disturb()
    outs = CUDA.Quantile_Multi(xs, q, None, hints)
    calls += 1
    for x, o, key in zip(xs, outs, keys):
        w = want(x, q)
        if not torch.equal(o, w):
            bad += 1
            print('MISMATCH round', r, 'key', key, 'got', o.tolist(), 'want', w.tolist(), flush=True)
    if jobs == 1:                                                            # the single-tensor entry points too
        disturb()
        o = CUDA.Quantile_Hinted(xs[0], q, hints[0])
        if not torch.equal(o, want(xs[0], q)): bad += 1; print('MISMATCH (single)', r, keys[0], flush=True)
torch.cuda.synchronize()
print(f'rounds {rounds} calls {calls} mismatches {bad} in {time.time() - t0:.1f} s (seed {seed})')
sys.exit(1 if bad else 0)
