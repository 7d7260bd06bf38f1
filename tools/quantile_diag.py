"""Developer aid: what the quantile filter did for every observed tensor of a real ResNet-50 percentile calibration
(which sides select A settled from the list / the tie count / left open for the exact passes, list lengths vs what was
needed, hint state), read back from the workspace of the last launch sequence.
    python tools/quantile_diag.py [batches]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ppq_amd import CUDA, harness, ffi, _lib
from ppq_amd import observer as obs
from ppq_amd.calibration import RuntimeCalibrationPass

dev = 'cuda:0'
batches_n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lay = (ctypes.c_int64 * 8)()
_lib.lib.ppqhip_quantile_debug_layout(lay)
PREF, WORDS, OFF_SEL, OFF_SPEC, OFF_TICK, OFF_TABLE, P_CNT, P_TIE = [int(v) for v in lay]

graph = harness.resnet50_graph(seed=0)
harness.quantize_graph(graph, 'percentile', hist_bins=2048)
ex = harness.TorchExecutor(graph, dev)
harness.ParameterQuantizePass().optimize(graph)
g = torch.Generator().manual_seed(0)
batches = [torch.rand(32, 3, 224, 224, generator=g).to(dev) for _ in range(batches_n)]

calls = []
orig = ffi.CUDA.Quantile_Multi


def spy(tensors, q, dests=None, hints=None):
    out = orig(tensors, q, dests, hints)
    torch.cuda.synchronize()
    ws = ffi._workspaces[(torch.device(dev).index, ffi._stream())]
    raw = ws.cpu().numpy().view(np.uint32)
    rows = []
    for j, t in enumerate(tensors):
        base = PREF // 4 + j * WORDS
        P = raw[base + OFF_SPEC: base + OFF_SPEC + 48]
        S = raw[base + OFF_SEL: base + OFF_SEL + 16]
        tick = raw[base + OFF_TICK: base + OFF_TICK + 3]
        n = t.numel()
        k_hi = int(np.rint(np.float32(n) * np.float32(q))); k_lo = int(np.rint(np.float32(n) * (np.float32(1) - np.float32(q))))
        k_hi = min(max(k_hi, 0), n - 1); k_lo = min(max(k_lo, 0), n - 1)
        h = hints[j].cpu().numpy().view(np.uint32) if hints and hints[j] is not None else None
        rows.append(dict(n=n, shape=tuple(t.shape), enabled=int(P[0]), hot=int(P[7]), ovf=(int(P[9]), int(P[10])),
                         cnt=(int(P[P_CNT:P_CNT + 8].sum()), int(P[P_CNT + 8:P_CNT + 16].sum())),
                         tie=(int(P[P_TIE:P_TIE + 8].sum()), int(P[P_TIE + 8:P_TIE + 16].sum())),
                         wanted=(n - k_hi, k_lo + 1), mode=(int(S[2]), int(S[10])), tick=tuple(int(v) for v in tick),
                         T=(hex(int(P[1])), hex(int(P[2]))), hint=None if h is None else (int(h[0]), int(h[2]), int(h[7]))))
    header = raw[:4].tolist()
    calls.append((header, rows))
    return out


obs.CUDA.Quantile_Multi = staticmethod(spy)
p = RuntimeCalibrationPass(method='percentile', check_steps=False)
p.optimize(graph, dataloader=batches, executor=ex, calib_steps=batches_n)
for ci, (header, rows) in enumerate(calls):
    f1 = sum(1 for r in rows if r['tick'][0]); f2 = sum(1 for r in rows if r['tick'][1]); f3 = sum(1 for r in rows if r['tick'][2])
    hot = sum(r['hot'] for r in rows)
    print(f'call {ci}: {len(rows)} jobs, header(cold, open, open3)={header[:3]}, hot jobs {hot}, jobs in F1/F2/F3: {f1}/{f2}/{f3}')
    if ci in (0, len(calls) - 1):
        for j, r in enumerate(rows):
            if r['tick'][0] or ci == len(calls) - 1 and j < 12:
                print('   ', j, r)
