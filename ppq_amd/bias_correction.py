"""BiasCorrectionPass -- mirror of ppq/quantization/optim/training.py:338-577.

    bias_error = reduce_mean(Y_fp32) - reduce_mean(Y_quant)      (per output channel)
    b <- b + bias_error, kept only when the block's MSE against the FP32 output does not get worse.

What runs where:
  * the per-channel DC term (``collect_bias``, training.py:438-448: ``torch.mean`` over every dim
    but the channel one) is one HIP reduction per forward, ``CUDA.ChannelMean`` ->
    ``ppqhip_channel_sum`` (double accumulation, fixed summation order);
  * the block forwards go through the executor (``partial_graph_forward``), i.e. through the HIP
    fake-quant kernels for every activated config;
  * blocks come from ppq_amd/blocks.py (the reference's TrainableBlock definition); ``block_size`` is the
    depth limit (default 1: every computing op is its own block, as in the reference's default).
"""
from collections import defaultdict
from typing import Callable, Iterable, List, Tuple

import torch

from .blocks import (PrefixCache, block_forward, collect, collect_all_fp_outputs, compute_block_loss, split_graph_into_blocks,
                     supports_prefix_cache, torch_mean_square_error)
from .calibration import QuantizationOptimizationPass
from .ffi import CUDA

BIAS_CORRECTION_INTERST_TYPE = {'Conv', 'ConvTranspose', 'Gemm'}      # ppq/core/common.py


def collect_bias(output: torch.Tensor, op_type: str) -> torch.Tensor:
    """training.py:438-448 -> [1, C] float32."""
    if output.ndim < 1: raise ValueError('Forward value has an unexpected dimension.')
    if op_type in {'Conv', 'ConvTranspose'}: axis = 1            # bias is added on axis 1
    elif op_type in {'Gemm'}: axis = output.ndim - 1             # bias is added on the last axis
    else: raise TypeError(f'Unsupported Operation type: {op_type}')
    return CUDA.ChannelMean(output, axis).unsqueeze(0)


class BiasCorrectionPass(QuantizationOptimizationPass):
    def __init__(self, interested_layers: List[str] = [], collecting_device: str = 'cuda',
                 steps: int = 32, block_size: int = 1) -> None:
        super().__init__(name='PPQ Bias Correction Pass')
        self.interested_layers = interested_layers
        self.steps = steps
        self.block_size = block_size
        self.collecting_device = collecting_device
        self.loss_fn = torch_mean_square_error
        self.report: List[Tuple[str, float, float]] = []

    # ---------------------------------------------------------------- one block (training.py:433-527)
    @ torch.no_grad()
    def correct_bias(self, qt_inputs, fp_outputs, block, executor, graph) -> Tuple[float, float]:
        pre_loss = compute_block_loss(block, qt_inputs, fp_outputs, executor, self.loss_fn)
        bias_cloned, interested_outputs = {}, []
        for op in block.rps:
            if (op.type in BIAS_CORRECTION_INTERST_TYPE and len(op.inputs) == 3 and op.inputs[-1].is_parameter
                    and isinstance(op.inputs[-1].value, torch.Tensor)):
                bias_cloned[op.name] = op.inputs[-1].value.clone()
                interested_outputs.append(op.outputs[0].name)
        if not interested_outputs: return pre_loss, pre_loss
        fp_cache, qt_cache = defaultdict(list), defaultdict(list)
        quantable = [op for op in block.rps if hasattr(op, 'config')]
        for op in quantable: op.dequantize()                              # phase 1: FP32 block outputs
        for qt_input in qt_inputs:
            outs = block_forward(executor, block.rps, qt_input, interested_outputs)
            for name, value in zip(interested_outputs, outs):
                fp_cache[name].append(collect_bias(value, graph.variables[name].source_op.type))
        for op in quantable: op.restore_quantize_state()                  # phase 2: quantised block outputs
        for qt_input in qt_inputs:
            outs = block_forward(executor, block.rps, qt_input, interested_outputs)
            for name, value in zip(interested_outputs, outs):
                qt_cache[name].append(collect_bias(value, graph.variables[name].source_op.type))
        for name in interested_outputs:
            if len(fp_cache[name]) == 0 or len(qt_cache[name]) == 0:
                raise ValueError('Bias correction failed, No data was collected.')
            DC_term_fp = torch.mean(torch.cat(fp_cache[name], dim=0), dim=0)
            DC_term_qt = torch.mean(torch.cat(qt_cache[name], dim=0), dim=0)
            bias = graph.variables[name].source_op.inputs[-1]
            bias.value += (DC_term_fp - DC_term_qt).to(bias.value.dtype)
        post_loss = compute_block_loss(block, qt_inputs, fp_outputs, executor, self.loss_fn)
        if post_loss > pre_loss:                                          # loss check: drop a worse result
            for op_name, value in bias_cloned.items(): graph.operations[op_name].inputs[-1].value.copy_(value)
            post_loss = pre_loss
        return pre_loss, post_loss

    def optimize(self, graph, dataloader: Iterable, executor, collate_fn: Callable = None, **kwargs) -> None:
        batches = []
        for data in dataloader:
            batches.append(collate_fn(data) if collate_fn is not None else data)
            if len(batches) >= self.steps: break
        self.report = []
        blocks = split_graph_into_blocks(graph, graph.topological_sort(), self.block_size,
                                         interested_layers=self.interested_layers)
        # FP32 targets of every block from ONE dequantised forward per batch (they depend on the parameters stored at quantisation
        # time only) and quantised block inputs computed incrementally (blocks.PrefixCache: a corrected block invalidates what it
        # feeds) -- the reference runs two full forwards per block and batch (training.py:224-298); same values
        all_fp = collect_all_fp_outputs(graph, blocks, executor, batches)
        prefix = PrefixCache(graph, executor, batches) if (all_fp is not None and supports_prefix_cache(executor)) else None
        for k, block in enumerate(blocks):
            targets = all_fp[k] if all_fp is not None else None
            if all_fp is not None: all_fp[k] = None
            if prefix is not None: qt_inputs, fp_outputs = prefix.inputs_of(block), targets
            else: qt_inputs, fp_outputs = collect(graph, block, executor, batches, fp_outputs=targets)
            pre_loss, post_loss = self.correct_bias(qt_inputs, fp_outputs, block, executor, graph)
            if prefix is not None: prefix.invalidate(block)
            self.report.append((block.sp.name, pre_loss, post_loss))
