"""Where the time of the plain drop-in seam goes: the UNMODIFIED reference -- its executor, RuntimeCalibrationPass
and observers -- with libppq_hip.so as its kernel extension (`install_into_ppq()`), ResNet-50 KL 2048 bins, next to this package's
pass in the same reference pipeline (`install_plugins_into_ppq()`).  Per stack and batch size:
  * wall time of the pass and of its four parts (collect min/max, render, collect histograms, render): the reference's
    `calibrate` / observer `render_quantization_config` wrapped with timers (this tool only);
  * cProfile of one pass: top functions by own time;
  * run under `rocprofv3 --kernel-trace` (tools/r6_seam.sh) the kernel trace is summed per category: this library, torch's
    reductions (the reference observers' value.min() / value.max()), MIOpen / rocBLAS, other torch kernels.
    python tools/seam_attribution.py [--batch 32] [--steps 20] [--stack kernels|fast|observers|pass] [--profile]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--stack', default='kernels')
ap.add_argument('--profile', action='store_true')
ap.add_argument('--method', default='kl')
args = ap.parse_args()

import bench  # noqa: E402
import ppq_amd  # noqa: E402
from ppq_amd import harness  # noqa: E402
from oracle import reference_import as RI  # noqa: E402

stage = bench.staged_reference()
assert stage is not None, 'oracle/_ref/ppq_stage is missing (python -c "import __graft_entry__ as g; g.build()" where /root/reference exists)'
RI.load(stage)
if args.stack == 'fast': ppq_amd.install_into_ppq(fast_observers=True)
elif args.stack == 'kernels': ppq_amd.install_into_ppq()
elif args.stack == 'observers': ppq_amd.install_plugins_into_ppq(observers=True)       # the reference's pass builds THIS package's observers from its OBSERVER_TABLE
else: ppq_amd.install_plugins_into_ppq(observers=False)
import ppq.lib as PFL  # noqa: E402
from ppq.quantization.optim import RuntimeCalibrationPass as RefPass  # noqa: E402
from ppq_amd.calibration import RuntimeCalibrationPass as OurPass  # noqa: E402

dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
batches = [torch.rand(args.batch, 3, 224, 224, device=dev, generator=g) for _ in range(min(args.steps, 8))]
parts = {}


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try: return fn(*a, **k)
        finally:
            torch.cuda.synchronize(); parts[name] = parts.get(name, 0.0) + time.perf_counter() - t0
    return wrapper


def one(profile=False):
    rg, rex = RI.quantize_reference_graph(RI.to_reference_graph(harness.resnet50_graph(seed=0)), dev, batches[0], bins=2048, method=args.method)
    if args.stack in ('kernels', 'fast', 'observers'):
        p = RefPass(method=args.method)
        calls = [0]
        orig_cal = p.calibrate

        def calibrate(*a, **k):
            calls[0] += 1
            return timed(f'collect_{calls[0]}', orig_cal)(*a, **k)
        p.calibrate = calibrate
        run = lambda: p.optimize(graph=rg, dataloader=batches, executor=rex, calib_steps=args.steps, collate_fn=None)   # noqa: E731
    else:
        p = OurPass(method=args.method, check_steps=False, use_hip_graph='auto' if args.batch < 16 else False)
        run = lambda: PFL.Pipeline([p]).optimize(graph=rg, dataloader=batches, executor=rex, calib_steps=args.steps, collate_fn=None, verbose=False)   # noqa: E731
    parts.clear()
    pr = cProfile.Profile() if profile else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if pr: pr.enable()
    run()
    if pr: pr.disable()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dt, dict(parts), pr


one()                                                   # warm: MIOpen find, allocator
best = None
for _ in range(2):
    dt, pp, _ = one()
    if best is None or dt < best[0]: best = (dt, pp)
dt, pp = best
n = args.steps * args.batch
print(f'[{args.stack}] batch {args.batch} x {args.steps} steps: {dt * 1e3:.1f} ms per pass = {dt / args.steps * 1e3:.2f} ms per step = {n / dt:.0f} samples/s')
if pp:
    coll = sum(v for k, v in pp.items() if k.startswith('collect'))
    print('   ' + '  '.join(f'{k} {v * 1e3:.1f} ms' for k, v in sorted(pp.items())) + f'   render + the rest {max(0.0, dt - coll) * 1e3:.1f} ms')
if args.profile:
    dt, pp, pr = one(profile=True)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
    print(f'cProfile of one pass ({dt * 1e3:.1f} ms under the profiler), by own time:')
    print('\n'.join(line[:170] for line in s.getvalue().splitlines()[4:40]))
