"""How fast is the UNMODIFIED reference (its IR, executor, observers and RuntimeCalibrationPass) when its kernels are
libppq_hip.so (ppq_amd.install_into_ppq()), on the bench's own workload?  Needs a GPU and an importable reference
(tools/stage_reference.py stages one for the GPU box).  Prints one JSON line; compare with `python bench.py`."""
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
    RI.load()
    ppq_amd.install_into_ppq()
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(1234)
    batches = [torch.rand(batch, 3, 224, 224, device=dev, generator=g) for _ in range(steps)]
    times = []
    for rep in range(repeats + 1):                      # the first pass warms MIOpen / the allocator and is dropped
        rg, rex = RI.quantize_reference_graph(RI.to_reference_graph(harness.resnet50_graph(seed=0)), dev, batches[0], bins=2048, method='kl')
        torch.cuda.synchronize(); t0 = time.perf_counter()
        RI.calibrate(rg, rex, batches, method='kl')
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    times = sorted(times[1:])
    med = times[len(times) // 2]
    print(json.dumps({'what': 'unmodified reference RuntimeCalibrationPass(kl, 2048 bins) on libppq_hip.so, ResNet-50, '
                              f'{steps} batches x {batch}', 'samples_per_s': round(batch * steps / med, 1),
                      'ms_per_step': round(med / steps * 1e3, 2), 'all_s': [round(t, 4) for t in times],
                      'scales': len(RI.activation_scales(rg))}))


if __name__ == '__main__':
    main(*(int(a) for a in sys.argv[1:]))
