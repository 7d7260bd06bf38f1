"""Developer aid: run the block-wise LSQ pass on the small CNN with the HIP-graph step under several settings and print
why a capture failed (first error + where).  `python tools/lsq_graph_debug.py`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd import harness  # noqa: E402
from ppq_amd.calibration import RuntimeCalibrationPass  # noqa: E402
from ppq_amd.lsq import LearnedStepSizePass  # noqa: E402

DEV = 'cuda'


def run(tag, **kw):
    LearnedStepSizePass._graph_broken, LearnedStepSizePass.graph_error = False, None
    mode = kw.pop('mode', 'global')
    graph = harness.small_cnn_graph(seed=5, width=16)
    harness.quantize_graph(graph, 'minmax')
    for op in graph.operations.values():
        for cfg, var in op.config_with_variable:
            if var.is_parameter and cfg.state.value == 1: cfg.num_of_bits, cfg.quant_min, cfg.quant_max = 4, -8, 7
    ex = harness.TorchExecutor(graph, DEV)
    harness.ParameterQuantizePass().optimize(graph)
    g = torch.Generator().manual_seed(7)
    batches = [torch.rand(8, 3, 24, 24, generator=g).to(DEV) for _ in range(4)]
    RuntimeCalibrationPass(check_steps=False).optimize(graph, dataloader=batches, executor=ex, calib_steps=4)
    p = LearnedStepSizePass(steps=6, lr=1e-3, **kw)
    p.capture_error_mode = mode
    p.optimize(graph, batches, ex)
    print(f'[{tag}] stats={p.stats}', flush=True)


if __name__ == '__main__':
    run('default')
    run('no-groups', group_weights=False)
    run('thread_local', mode='thread_local')
    run('relaxed', mode='relaxed')
