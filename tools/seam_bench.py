"""Calibration samples/s of the three ways this library sits inside an importable, UNMODIFIED reference, on the bench's own
workload (ResNet-50, KL 2048 bins).  Needs a GPU and a staged reference.  One JSON line per stack.
  kernels : reference executor + reference pass + reference observers, kernels = libppq_hip.so   (install_into_ppq)
  observers: reference executor + reference pass, observers from this package                      (install_plugins_into_ppq)
  pass    : reference executor, this package's pass + observers in the reference's Pipeline
  harness : this package's executor + pass (what bench.py's headline measures)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reference_import as RI  # noqa: E402


def main(batch=32, steps=8, repeats=3):
    import ppq_amd
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass as OurPass
    RI.load()
    ppq_amd.install_into_ppq()
    import ppq.lib as PFL
    from ppq.quantization.optim import RuntimeCalibrationPass as RefPass
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(1234)
    batches = [torch.rand(batch, 3, 224, 224, device=dev, generator=g) for _ in range(steps)]

    def timed(stack):
        times = []
        for rep in range(repeats + 1):                      # the first pass warms MIOpen / the allocator and is dropped
            if stack == 'harness':
                hg = harness.resnet50_graph(seed=0)
                harness.quantize_graph(hg, 'kl', hist_bins=2048)
                ex = harness.TorchExecutor(hg, dev)
                harness.ParameterQuantizePass().optimize(hg)
                p, kw, graph = OurPass(method='kl', check_steps=False), {}, hg
            else:
                graph, ex = RI.quantize_reference_graph(RI.to_reference_graph(harness.resnet50_graph(seed=0)), dev, batches[0], bins=2048, method='kl')
                p = OurPass(method='kl', check_steps=False) if stack == 'pass' else RefPass(method='kl')
                kw = {'collate_fn': None}
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if stack == 'pass': PFL.Pipeline([p]).optimize(graph=graph, dataloader=batches, executor=ex, calib_steps=max(8, steps), verbose=False, **kw)
            else: p.optimize(graph=graph, dataloader=batches, executor=ex, calib_steps=max(8, steps), **kw)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        times = sorted(times[1:])
        med = times[len(times) // 2]
        n = max(8, steps)
        return {'stack': stack, 'samples_per_s': round(batch * n / med, 1), 'ms_per_step': round(med / n * 1e3, 2), 'all_s': [round(t, 4) for t in times]}
    print(json.dumps(timed('kernels')), flush=True)
    ppq_amd.install_plugins_into_ppq(observers=False)
    print(json.dumps(timed('pass')), flush=True)
    print(json.dumps(timed('harness')), flush=True)
    ppq_amd.install_plugins_into_ppq()
    print(json.dumps(timed('observers')), flush=True)


if __name__ == '__main__':
    main(*(int(a) for a in sys.argv[1:]))
