"""BiasCorrectionPass -- mirror of ppq/quantization/optim/training.py:338-577.

    bias_error = reduce_mean(Y_fp32) - reduce_mean(Y_quant)      (per output channel)
    b <- b + bias_error, kept only when the block's MSE against the FP32 output does not get worse.

What runs where:
  * the per-channel DC term (``collect_bias``, training.py:438-448: ``torch.mean`` over every dim
    but the channel one) is one HIP reduction per forward, ``CUDA.ChannelMean`` ->
    ``ppqhip_channel_sum`` (double accumulation, fixed summation order);
  * the block forwards go through the executor (``partial_graph_forward``), i.e. through the HIP
    fake-quant kernels for every activated config;
  * blocks are the reference's ``block_size = 1`` case: every Conv / ConvTranspose / Gemm that owns a
    bias parameter is its own block (the BlockBuilder graph search of training.py:191-222 for larger
    blocks is graph plumbing outside this package's scope).
"""
from collections import defaultdict
from typing import Callable, Dict, Iterable, List, Tuple

import torch

from .calibration import QuantizationOptimizationPass
from .ffi import CUDA

BIAS_CORRECTION_INTERST_TYPE = {'Conv', 'ConvTranspose', 'Gemm'}      # ppq/core/common.py


def torch_mean_square_error(y_pred: torch.Tensor, y_real: torch.Tensor) -> torch.Tensor:
    """ppq/quantization/measure/norm.py: mean over the batch of the per-sample mean squared error."""
    return torch.mean(torch.mean(torch.square(y_pred.flatten(1) - y_real.flatten(1)), dim=-1))


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
        if block_size != 1:
            raise NotImplementedError('BiasCorrectionPass: only block_size = 1 (one computing op per block)')
        self.interested_layers = interested_layers
        self.steps = steps
        self.block_size = block_size
        self.collecting_device = collecting_device
        self.loss_fn = torch_mean_square_error
        self.report: List[Tuple[str, float, float]] = []

    # ---------------------------------------------------------------- data collection (training.py:224-298)
    def collect(self, graph, op, executor, batches) -> Tuple[List[Dict[str, torch.Tensor]], List[Dict[str, torch.Tensor]]]:
        feeds = [v for v in op.inputs if not v.is_parameter]
        quantable = [o for o in graph.operations.values() if hasattr(o, 'config')]
        for o in quantable: o.dequantize()
        fp_outputs = [{op.outputs[0].name: executor.forward(b, [op.outputs[0].name])[0]} for b in batches]
        for o in quantable: o.restore_quantize_state()
        qt_inputs = []
        for b in batches:
            if all(v.name in graph.inputs for v in feeds):
                vals = [b if isinstance(b, torch.Tensor) else b[v.name] for v in feeds]
            else:
                vals = executor.forward(b, [v.name for v in feeds])
            qt_inputs.append({v.name: x for v, x in zip(feeds, vals)})
        return qt_inputs, fp_outputs

    def compute_block_loss(self, op, qt_inputs, fp_outputs, executor) -> float:
        """training.py:300-335."""
        name, loss = op.outputs[0].name, 0.0
        for qt_input, fp_output in zip(qt_inputs, fp_outputs):
            out = executor.partial_graph_forward([op], qt_input, [name])[0]
            loss += float(self.loss_fn(out, fp_output[name]))
        return loss / len(qt_inputs)

    # ---------------------------------------------------------------- one block (training.py:433-527)
    @ torch.no_grad()
    def correct_bias(self, qt_inputs, fp_outputs, op, executor) -> Tuple[float, float]:
        pre_loss = self.compute_block_loss(op, qt_inputs, fp_outputs, executor)
        bias = op.inputs[-1]
        bias_cloned = bias.value.clone()
        name = op.outputs[0].name
        fp_cache, qt_cache = defaultdict(list), defaultdict(list)
        op.dequantize()                                                   # phase 1: FP32 block output
        for qt_input in qt_inputs:
            out = executor.partial_graph_forward([op], qt_input, [name])[0]
            fp_cache[name].append(collect_bias(out, op.type))
        op.restore_quantize_state()                                       # phase 2: quantised block output
        for qt_input in qt_inputs:
            out = executor.partial_graph_forward([op], qt_input, [name])[0]
            qt_cache[name].append(collect_bias(out, op.type))
        if len(fp_cache[name]) == 0 or len(qt_cache[name]) == 0:
            raise ValueError('Bias correction failed, No data was collected.')
        DC_term_fp = torch.mean(torch.cat(fp_cache[name], dim=0), dim=0)
        DC_term_qt = torch.mean(torch.cat(qt_cache[name], dim=0), dim=0)
        bias.value += (DC_term_fp - DC_term_qt).to(bias.value.dtype)
        post_loss = self.compute_block_loss(op, qt_inputs, fp_outputs, executor)
        if post_loss > pre_loss:                                          # loss check: drop a worse result
            bias.value.copy_(bias_cloned)
            post_loss = pre_loss
        return pre_loss, post_loss

    def optimize(self, graph, dataloader: Iterable, executor, collate_fn: Callable = None, **kwargs) -> None:
        batches = []
        for data in dataloader:
            batches.append(collate_fn(data) if collate_fn is not None else data)
            if len(batches) >= self.steps: break
        self.report = []
        for op in graph.topological_sort():
            if op.type not in BIAS_CORRECTION_INTERST_TYPE or not hasattr(op, 'config'): continue
            if self.interested_layers and op.name not in self.interested_layers: continue
            if not (len(op.inputs) == 3 and op.inputs[-1].is_parameter
                    and isinstance(op.inputs[-1].value, torch.Tensor)): continue   # no bias: skipped
            qt_inputs, fp_outputs = self.collect(graph, op, executor, batches)
            pre_loss, post_loss = self.correct_bias(qt_inputs, fp_outputs, op, executor)
            self.report.append((op.name, pre_loss, post_loss))
