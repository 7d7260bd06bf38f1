import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppq_amd import harness
from ppq_amd.calibration import RuntimeCalibrationPass
from ppq_amd.lsq import LearnedStepSizePass
from ppq_amd.blocks import split_graph_into_blocks, collect
DEV='cuda'
graph = harness.yolov6s_graph(seed=3)
harness.quantize_graph(graph, 'minmax')
for op in graph.operations.values():
    for cfg, var in op.config_with_variable:
        if var.is_parameter and cfg.state.value == 1:
            cfg.num_of_bits, cfg.quant_min, cfg.quant_max = 4, -8, 7
ex = harness.TorchExecutor(graph, DEV)
harness.ParameterQuantizePass().optimize(graph)
g = torch.Generator().manual_seed(9)
batches = [torch.rand(2, 3, 160, 160, generator=g).to(DEV) for _ in range(8)]
RuntimeCalibrationPass().optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
lsq = LearnedStepSizePass(steps=40, lr=1e-4, block_size=5)
names=[op.outputs[0].name for op in graph.operations.values()]
for block in split_graph_into_blocks(graph, graph.topological_sort(), 5):
    qt_inputs, fp_outputs = collect(graph, block, ex, batches)
    qi=max(float(v.abs().max()) for d in qt_inputs for v in d.values()); fo=max(float(v.abs().max()) for d in fp_outputs for v in d.values())
    pre, post = lsq.finetune(block, ex, qt_inputs, fp_outputs)
    bad=[v.name for v in graph.variables.values() if v.is_parameter and not torch.isfinite(v.value).all()]
    sc=[(op.name, vv.name, float(c.scale.abs().max()), float(c.scale.abs().min())) for op in block.rps for c,vv in op.config_with_variable if c.scale is not None and (not torch.isfinite(c.scale).all() or float(c.scale.abs().max())>1e3 or float(c.scale.min())<=0)]
    print(f'{str(block):50s} in max {qi:10.4g} fp out max {fo:10.4g} pre {pre:12.5g} post {post:12.5g} bad params {bad[:3]} odd scales {sc[:3]}', flush=True)
