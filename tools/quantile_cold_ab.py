"""A hinted single-tensor quantile call that cannot use its hint: `fresh` = a hint address the library meets for the first time (the
general sequence), `cold` = a hint zeroed in place (the exact passes of the select launch), `hot` = settled from the hint.  Event-timed,
100 calls per line.    python tools/quantile_cold_ab.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from ppq_amd import CUDA
from ppq_amd.ffi import quantile_hint
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
keep = []
for m in (1, 4, 8, 16, 32):
    n = m * 512 * 56 * 56
    rot = max(2, min(8, (1 << 30) // (4 * n)))
    xs = [torch.randn(n, device=dev, generator=g) for _ in range(rot)]
    hint = quantile_hint(dev)
    fresh = [quantile_hint(dev) for _ in range(200)]; keep.append(fresh)      # kept: a freed hint's address would come back as a MET one
    for mode in ('fresh', 'cold', 'hot'):
        for it in range(2):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(100):
                if mode == 'cold': hint.zero_()
                CUDA.Quantile_Hinted(xs[i % rot], 0.9999, fresh[it * 100 + i] if mode == 'fresh' else hint)
            e1.record(); torch.cuda.synchronize()
        print(f'x{m} {mode}: {e0.elapsed_time(e1) * 10:.1f} us per call (incl. the 1.5 us memset when cold)', flush=True)
