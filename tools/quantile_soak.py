"""Soak test of quantile.hip on a GPU: random job mixes, sizes, distributions, q and hint states, every result compared with
a device sort (the reference's index rule, sort.cu:13-19) -- while a second stream keeps the chip unevenly busy, which is
where a missing release / acquire in the exact passes' "last workgroup" tails would show (MI355X_MICROARCH.md: test every
hand-off under uneven load).
    python tools/quantile_soak.py [rounds] [seed] [big | single]
`single`: only the single-tensor entry point with an observer's hint (quantile.hip "ONE hinted tensor: two launches"): sizes on every
geometry of its filter, slots of every length (a hot channel fills one workgroup's slot, an outlier burst overflows it), stale, garbage and
fresh hints, two streams at once."""
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
big = len(sys.argv) > 3 and sys.argv[3] == 'big'
single = len(sys.argv) > 3 and sys.argv[3] == 'single'        # also 25 M / 51 M element tensors (the sharded lists, many slices per job)
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
if single:
    other = torch.cuda.Stream()
    for r in range(rounds):
        n = int(rng.choice([262_144, 300_001, 1_605_632, 1_605_635, 3_211_264, 4_194_304, 4_194_312, 6_422_528, 12_845_056 + 3, 25_690_112]))
        kind = int(rng.integers(0, 11))
        q = float(rng.choice([0.9999, 0.9999, 0.999, 0.99]))
        key = (kind, n, q)
        for b in range(int(rng.integers(2, 6))):
            if kind < 8: x = make(n, kind)
            elif kind == 8:                   # a hot channel: ~3000 consecutive elements far above everything else (one workgroup's slot fills up)
                x = torch.randn(n, device=dev, generator=g); a0 = int(rng.integers(0, n - 4000)); x[a0:a0 + 3000] = 50 + torch.rand(3000, device=dev, generator=g)
            elif kind == 9:                   # [N, C, H, W]-like: every 37th run of 3136 elements is 25x larger
                x = torch.randn(n, device=dev, generator=g); v = x[: (n // 3136) * 3136].view(-1, 3136); v[::37] *= 25
            else:                             # an outlier burst longer than a workgroup can stage
                x = torch.randn(n, device=dev, generator=g); a0 = int(rng.integers(0, n - 9000)); x[a0:a0 + 8000] = -1e6 - torch.rand(8000, device=dev, generator=g)
            mode = rng.random()
            if mode < 0.8: h = streams.setdefault(key, quantile_hint(dev))
            elif mode < 0.9: h = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, 8, dtype=np.int64).astype(np.int32)).to(dev)
            else:                                # claims to be valid, thresholds from another distribution
                qf = np.float32(q); k_hi = int(np.rint(np.float32(n) * qf)); k_lo = int(np.rint(np.float32(n) * (np.float32(1) - qf)))
                t_hi = int(rng.integers(0x80000000, 0xC2000000)); t_lo = int(rng.integers(0x3D000000, 0x7FFFFFFF))
                h = torch.tensor([1, t_hi - (1 << 32), 1, t_lo, n, k_hi, k_lo, 0], dtype=torch.int64).to(torch.int32).to(dev)
            w = want(x, q)
            if not os.environ.get('SOAK_NODISTURB'): disturb()
            two = rng.random() < 0.3 and not os.environ.get('SOAK_NOTWO')
            if two:
                h2 = quantile_hint(dev) if rng.random() < 0.5 else streams.setdefault((kind, n, q, 'b'), quantile_hint(dev))
                other.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(other): o2 = CUDA.Quantile_Hinted(x, q, h2)
            if os.environ.get('SOAK_TIMES'): torch.cuda.synchronize(); tc = time.time()
            o = CUDA.Quantile_Hinted(x, q, h)
            if os.environ.get('SOAK_TIMES'):
                torch.cuda.synchronize(); tc = time.time() - tc
                if tc > 0.02: print(f'SLOW {tc * 1e3:.1f} ms: round {r} key {key} batch {b} mode {mode:.2f} two {two} hint {h.cpu().tolist()}', flush=True)
            calls += 1
            if two:
                torch.cuda.current_stream().wait_stream(other)
                if not torch.equal(o2, w): bad += 1; print('MISMATCH (second stream)', r, key, o2.tolist(), w.tolist(), flush=True)
            if not torch.equal(o, w): bad += 1; print('MISMATCH', r, key, b, o.tolist(), w.tolist(), h.cpu().tolist(), flush=True)
    torch.cuda.synchronize()
    used = sum(int(h.cpu()[7]) for h in streams.values())
    print(f'single: rounds {rounds} calls {calls} settled from a hint {used} mismatches {bad} in {time.time() - t0:.1f} s (seed {seed})')
    sys.exit(1 if bad else 0)
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
