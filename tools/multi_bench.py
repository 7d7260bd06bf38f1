"""Device time of the multi-tensor observer launches on a ResNet-50-like set of activation tensors
(batch 32): ONE launch over all of them vs one launch per tensor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppq_amd import CUDA
dev = 'cuda'
B = 32
shapes = [(B, 64, 112, 112)] + [(B, 256, 56, 56)] * 3 + [(B, 64, 56, 56)] * 6 + [(B, 512, 28, 28)] * 4 + [(B, 128, 28, 28)] * 8 \
    + [(B, 1024, 14, 14)] * 6 + [(B, 256, 14, 14)] * 12 + [(B, 2048, 7, 7)] * 3 + [(B, 512, 7, 7)] * 6 + [(B, 1000)]
torch.manual_seed(0)
xs = [torch.relu(torch.randn(*s, device=dev)) for s in shapes]
total = sum(x.numel() for x in xs) * 4
R, S = CUDA.hist_rows(), CUDA.minmax_slots()
rows = [torch.zeros(R, 2048, dtype=torch.int32, device=dev) for _ in xs]
seed = torch.tensor([float('inf'), float('-inf')], device=dev)
slots = [seed.repeat(S, 1).contiguous() for _ in xs]
scales = [float(x.max()) / 2048 for x in xs]
from ppq_amd.ffi import quantile_hint
qhints = [quantile_hint(x.device) for x in xs]


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


def per_tensor_hist():
    for x, r, s in zip(xs, rows, scales): CUDA.Histogram_T_Rows(x, r, s)


def per_tensor_minmax():
    for x, sl in zip(xs, slots): CUDA.MinMax_T_Slots(x, sl)


def per_tensor_quantile():
    for x in xs: CUDA.Quantile(x, 0.9999)


print(f'{len(xs)} tensors, {total / 1e6:.0f} MB')
for name, fn in (('hist multi', lambda: CUDA.Histogram_T_Rows_Multi(xs, rows, scales)), ('hist per-tensor', per_tensor_hist),
                 ('minmax multi', lambda: CUDA.MinMax_T_Slots_Multi(xs, slots)), ('minmax per-tensor', per_tensor_minmax),
                 ('quantile multi (cold)', lambda: CUDA.Quantile_Multi(xs, 0.9999)),
                 ('quantile multi (hinted)', lambda: CUDA.Quantile_Multi(xs, 0.9999, None, qhints)),
                 ('quantile per-tensor', per_tensor_quantile)):
    us = timed(fn)
    print(f'{name:18s} {us:9.1f} us  {total / us / 1e6:7.2f} TB/s')

# ---- the weights of ResNet-50 (54 tensors, 0.01 .. 9 MB): ONE fake-quant launch / ONE min-max launch for all of them
from ppq_amd import harness  # noqa: E402
from ppq_amd.ffi import LinearQuantizePlan  # noqa: E402
g = harness.resnet50_graph(seed=0)
ws = [v.value.to(dev) for op in g.operations.values() for i, v in enumerate(op.inputs) if v.is_parameter and i == 1]
wbytes = sum(w.numel() for w in ws) * 4
scs = [torch.rand(w.shape[0], device=dev) * 0.01 + 0.001 for w in ws]
ofs = [torch.zeros(w.shape[0], device=dev) for w in ws]
plans = [LinearQuantizePlan([(w.clone() if k else w, s, o, 0, -128, 127) for w, s, o in zip(ws, scs, ofs)]) for k in range(4)]   # rotate 4 arenas
mins = [torch.empty(w.shape[0], device=dev) for w in ws]; maxs = [torch.empty(w.shape[0], device=dev) for w in ws]
k = [0]


def fq_multi():
    k[0] += 1
    plans[k[0] % 4].run()


def fq_per_tensor():
    for w, s, o in zip(ws, scs, ofs): CUDA.LinearQuantize_C(w, s, o, 0, -128, 127, 0)


def mm_per_tensor():
    for w, a, b in zip(ws, mins, maxs):
        a.fill_(float('inf')); b.fill_(float('-inf'))
        CUDA.MinMax_C(w, 0, a, b)


print(f'{len(ws)} weights, {wbytes / 1e6:.0f} MB')
for name, fn, nbytes in (('fq_linear multi', fq_multi, 2 * wbytes), ('fq_linear per-tensor', fq_per_tensor, 2 * wbytes),
                         ('minmax_c multi (fresh)', lambda: CUDA.MinMax_C_Multi(ws, [0] * len(ws), mins, maxs, True), wbytes),
                         ('minmax_c per-tensor (+2 fills)', mm_per_tensor, wbytes)):
    us = timed(fn)
    print(f'{name:32s} {us:9.1f} us  {nbytes / us / 1e6:7.2f} TB/s')

