"""Diagnostic: is a full quantised forward of the YOLOv6-s-like graph bitwise repeatable, and does PrefixCache differ from it
by more than two full forwards differ from each other?   python tools/prefix_diag.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd import harness  # noqa: E402
from ppq_amd.blocks import PrefixCache, collect, split_graph_into_blocks  # noqa: E402
from ppq_amd.calibration import RuntimeCalibrationPass  # noqa: E402

DEV = 'cuda'
if len(sys.argv) > 1 and sys.argv[1] == 'det':
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    print('deterministic mode')
graph = harness.yolov6s_graph(seed=1)
harness.quantize_graph(graph, 'minmax')
for op in graph.operations.values():
    for cfg, var in op.config_with_variable:
        if var.is_parameter and cfg.state.value == 1: cfg.num_of_bits, cfg.quant_min, cfg.quant_max = 4, -8, 7
ex = harness.TorchExecutor(graph, DEV)
harness.ParameterQuantizePass().optimize(graph)
g = torch.Generator().manual_seed(3)
batches = [torch.rand(2, 3, 96, 96, generator=g).to(DEV) for _ in range(2)]
RuntimeCalibrationPass(check_steps=False).optimize(graph, dataloader=batches, executor=ex, calib_steps=2)
blocks = split_graph_into_blocks(graph, graph.topological_sort(), 5)
for trial in range(5):
    prefix = PrefixCache(graph, ex, batches)
    worst_noise = worst_diff = 0.0
    for k, block in enumerate(blocks):
        got = prefix.inputs_of(block)
        a, _ = collect(graph, block, ex, batches, fp_outputs=[{} for _ in batches])
        b, _ = collect(graph, block, ex, batches, fp_outputs=[{} for _ in batches])
        for x, y, z in zip(got, a, b):
            for n in x:
                step = float(y[n].abs().max()) / 127.0 + 1e-12
                noise = float((y[n] - z[n]).abs().max()) / step
                diff = float((x[n] - y[n]).abs().max()) / step
                fn = float(((y[n] - z[n]).abs() > 0).float().mean()); fd = float(((x[n] - y[n]).abs() > 0).float().mean())
                worst_noise, worst_diff = max(worst_noise, noise), max(worst_diff, diff)
                if False: print(f'trial {trial} block {k} {n}: full-vs-full {noise:.2f} steps ({fn:.4f}), cache-vs-full {diff:.2f} steps ({fd:.4f})')
        with torch.no_grad():
            for op in block.rps:
                for v in op.inputs:
                    if v.is_parameter and isinstance(v.value, torch.Tensor) and v.value.dim() == 4: v.value.mul_(1.0 + 0.01 * (k % 3))
        prefix.invalidate(block)
    print(f'trial {trial}: worst full-vs-full {worst_noise:.2f} steps, worst cache-vs-full {worst_diff:.2f} steps', flush=True)
