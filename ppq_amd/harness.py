"""Measurement harness: the smallest caller that can drive the calibration hot path end to end
when PPQ itself is not importable (the GPU box has no /root/reference).

NOT a re-implementation of PPQ's graph IR / executor / quantizers (all out of scope, SURVEY
section 2): it offers just the slice of their interfaces that ``RuntimeCalibrationPass`` and the
observers touch, with the reference's names and call protocol --

* ``Variable`` / ``Operation`` / ``QuantableOperation`` / ``BaseGraph``   ppq/IR/base/graph.py:15-260,
                                                                           ppq/IR/quantize.py:15-140
* ``TorchExecutor.forward(inputs, output_names, hooks)``                   ppq/executor/torch.py:365-577
  (quantize inputs -> pre_forward_hook -> op -> quantize outputs -> post_forward_hook)
* a TensorRT-style INT8 policy (``quantize_graph``)                        quantizer/TensorRTQuantizer.py
  + the state edits of QuantizeFusionPass / QuantizeSimplifyPass           optim/refine.py
* ``ParameterQuantizePass`` / ``ParameterBakingPass``                      now ppq_amd/parameters.py (re-exported here)
* ``resnet50_graph`` -- the ResNet-50 topology (53 Conv + 1 Gemm, BN pre-folded) with seeded
  He-initialised weights, standing in for the ONNX model that cannot be loaded here (no `onnx`).

The dense math (conv / gemm) is PyTorch-ROCm (MIOpen / rocBLAS), exactly as in the reference; only
the quantization simulation goes through this package's kernels.
"""
from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F

from .core import LinearQuantizationConfig, QuantizationStates, TensorQuantizationConfig, is_initial
from .core import QuantizationProperty as P
from .qfunction import PPQuantFunction

PASSIVE_OPERATIONS = {'MaxPool', 'GlobalMaxPool', 'Reshape', 'Flatten', 'Identity', 'Dropout', 'Slice', 'Pad', 'Resize',
                      'Split', 'Transpose', 'Interp', 'Squeeze', 'Unsqueeze'}        # ppq/core/common.py:50-53
COMPUTING_OP = {'Conv', 'Gemm', 'ConvTranspose', 'MatMul'}


class Variable:
    def __init__(self, name: str, value: torch.Tensor = None, is_parameter: bool = False):
        self.name = name
        self.value = value
        self.stored_value = None         # QuantableVariable.stored_value, IR/quantize.py:183-205 (set when its op is quantised)
        self.is_parameter = is_parameter
        self.dest_ops: List['Operation'] = []
        self.source_op: Optional['Operation'] = None

    def __hash__(self): return hash(self.name)
    def __eq__(self, o): return isinstance(o, Variable) and o.name == self.name


class OperationQuantizationConfig:
    """ppq/core/quant.py:952-1013."""
    def __init__(self, input_quantization_configs, output_quantization_configs):
        self.input_quantization_config: List[TensorQuantizationConfig] = input_quantization_configs
        self.output_quantization_config: List[TensorQuantizationConfig] = output_quantization_configs


class Operation:
    def __init__(self, name: str, op_type: str, attributes: dict = None, inputs=None, outputs=None):
        self.name, self.type = name, op_type
        self.attributes = attributes or {}
        self.inputs: List[Variable] = inputs or []
        self.outputs: List[Variable] = outputs or []

    @ property
    def parameters(self) -> List[Variable]:
        return [v for v in self.inputs if v.is_parameter]

    def __str__(self): return f'{self.name}({self.type})'


class QuantableOperation(Operation):
    """ppq/IR/quantize.py:15-140."""
    def __init__(self, op: Operation, config: OperationQuantizationConfig):
        super().__init__(op.name, op.type, op.attributes, op.inputs, op.outputs)
        self.config = config
        self.store_parameter_value()

    def store_parameter_value(self):
        """IR/quantize.py:113-122: keep a copy of every parameter as it is when the operation is quantised."""
        for var in self.inputs:
            if var.is_parameter and isinstance(var.value, torch.Tensor): var.stored_value = var.value.detach().clone()
        return self

    def _swap_parameters(self) -> None:
        for var in self.inputs:
            if var.is_parameter and isinstance(var.value, torch.Tensor) and var.stored_value is not None:
                parked = var.value
                var.value = var.stored_value.to(parked.device)
                var.stored_value = parked

    @ property
    def config_with_variable(self):
        return list(zip(self.config.input_quantization_config, self.inputs)) + \
            list(zip(self.config.output_quantization_config, self.outputs))

    def dequantize(self):
        """ppq/IR/quantize.py:124-141: park every config in FP32, remembering its state, AND swap every parameter with
        its stored value -- a dequantised operation computes with the parameters it had when it was quantised (the
        current, possibly finetuned / bias-corrected / baked ones wait in stored_value).  This is what makes the FP32
        targets of the training based passes independent of what earlier blocks did to the parameters."""
        if getattr(self, '_dequantized', False): return self
        for cfg, _ in self.config_with_variable:
            cfg.detail['Stored State'] = cfg.state
            cfg.state = QuantizationStates.FP32
        self._swap_parameters()
        self._dequantized = True
        return self

    def restore_quantize_state(self):
        """ppq/IR/quantize.py:143-160: the states and the current parameters come back."""
        if not getattr(self, '_dequantized', False): return self
        for cfg, _ in self.config_with_variable:
            if 'Stored State' in cfg.detail: cfg.state = cfg.detail.pop('Stored State')
        self._swap_parameters()
        self._dequantized = False
        return self

    def baking_parameters(self, quant_func: Callable):
        """IR/quantize.py:98-111."""
        for config, var in self.config_with_variable:
            if var.is_parameter and QuantizationStates.is_activated(config.state):
                var.value = quant_func(var.value, config)
                config.state = (QuantizationStates.BAKED if config.state == QuantizationStates.ACTIVATED
                                else QuantizationStates.PASSIVE_BAKED)


class BaseGraph:
    def __init__(self, name: str):
        self.name = name
        self.operations: Dict[str, Operation] = {}
        self.variables: Dict[str, Variable] = {}
        self.inputs: Dict[str, Variable] = {}
        self.outputs: Dict[str, Variable] = {}

    def create_variable(self, name: str, value=None, is_parameter=False) -> Variable:
        v = Variable(name, value, is_parameter)
        self.variables[name] = v
        return v

    def create_operation(self, op_type: str, name: str, inputs: List[Variable], attributes: dict = None) -> Variable:
        out = self.create_variable(name + '_out')
        op = Operation(name, op_type, attributes, list(inputs), [out])
        out.source_op = op
        for v in inputs: v.dest_ops.append(op)
        self.operations[name] = op
        return out

    def topological_sort(self) -> List[Operation]:
        return list(self.operations.values())        # operations are created in execution order


# ------------------------------------------------------------------------------------ op library
def _forward(op: Operation, x: List[torch.Tensor]):
    t, a = op.type, op.attributes
    if t == 'Conv':
        return F.conv2d(x[0], x[1], x[2] if len(x) > 2 else None, stride=a.get('strides', 1),
                        padding=a.get('pads', 0), groups=a.get('group', 1))
    if t == 'Gemm': return F.linear(x[0], x[1], x[2] if len(x) > 2 else None)
    if t == 'MatMul': return torch.matmul(x[0], x[1])
    if t == 'Relu': return F.relu(x[0])
    if t == 'Gelu': return F.gelu(x[0])
    if t == 'Add': return x[0] + x[1]
    if t == 'MaxPool': return F.max_pool2d(x[0], a['kernel_shape'], a.get('strides', 1), a.get('pads', 0))
    if t == 'GlobalAveragePool': return F.adaptive_avg_pool2d(x[0], 1)
    if t == 'Flatten': return torch.flatten(x[0], 1)
    if t == 'LayerNormalization': return F.layer_norm(x[0], x[0].shape[-1:], x[1], x[2])
    if t == 'Softmax': return F.softmax(x[0], dim=a.get('axis', -1))
    if t == 'Mul': return x[0] * (x[1] if len(x) > 1 else a['value'])
    if t == 'Reshape': return x[0].reshape(a['shape'])
    if t == 'Transpose': return x[0].permute(a['perm'])
    if t == 'Concat': return torch.cat([v.expand(x[-1].shape[0], *v.shape[1:]) if v.shape[0] == 1 else v for v in x],
                                       dim=a.get('axis', 1))
    if t == 'Slice': return x[0].narrow(a['axis'], a['start'], a['length'])
    if t == 'Resize': return F.interpolate(x[0], scale_factor=a.get('scale', 2), mode=a.get('mode', 'nearest'))
    raise NotImplementedError(f'Graph op: {op.name}({op.type}) has no backend implementation')


class TorchExecutor:
    """The forward loop of ppq/executor/torch.py:457-577 (hook protocol of executor/base.py:44-102)."""
    def __init__(self, graph: BaseGraph, device: str = 'cuda'):
        self._graph = graph
        self._device = device
        self._default_quant_fn = PPQuantFunction
        self.cache_parameter_quantization = False     # opt-in: see _quantize_parameter
        self._param_cache: Dict[str, tuple] = {}
        self.fuse_parameter_quantization = True       # all weights of a forward in ONE launch: _fused_parameters
        self._plans: list = []
        self._plan_signature = None
        self._plan_cache: Dict[tuple, list] = {}
        self._fused: Dict[tuple, torch.Tensor] = {}
        self._delegates: Dict[object, Callable] = {}
        self.channels_last = False                    # see use_channels_last()
        for v in graph.variables.values():
            if v.is_parameter and v.value is not None: v.value = v.value.to(device)

    def use_channels_last(self) -> 'TorchExecutor':
        """Keep 4-D activations and conv weights in channels-last memory format: MIOpen's FP32 convolutions
        are ~8 % faster in NHWC on MI355X (tools/conv_format_probe.py) and the per-tensor statistics /
        fake-quant kernels stream any dense layout in storage order (ffi._dense), so no layout copy appears
        anywhere on the calibration path.  Values are those of the NCHW run up to the convolution
        algorithm's own rounding."""
        self.channels_last = True
        for v in self._graph.variables.values():
            if v.is_parameter and isinstance(v.value, torch.Tensor) and v.value.dim() == 4:
                v.value = v.value.contiguous(memory_format=torch.channels_last)
        return self

    def _place(self, value: torch.Tensor) -> torch.Tensor:
        value = value.to(self._device)
        if self.channels_last and value.dim() == 4: value = value.contiguous(memory_format=torch.channels_last)
        return value

    def register_quantize_delegate(self, config, delegator: Callable) -> None:
        """ppq/executor/torch.py:296-323: a delegator takes over the quantisation of one config."""
        self._delegates[config] = delegator

    def remove_quantize_delegate(self, config) -> None:
        self._delegates.pop(config, None)

    def quantize_function(self, tensor: torch.Tensor, config=None) -> torch.Tensor:
        if self._delegates and config in self._delegates:          # torch.py:610-613
            return self._delegates[config](tensor, config)
        return self._default_quant_fn(tensor, config)

    def forward_with_gradient(self, inputs, output_names: List[str] = None, hooks=None):
        """torch.py:412-455: the same loop with autograd enabled (finetuning passes)."""
        return TorchExecutor.forward.__wrapped__(self, inputs, output_names, hooks)

    def partial_graph_forward(self, operations: List[Operation], feed_dict: Dict[str, torch.Tensor],
                              output_names: List[str], with_gradient: bool = False) -> List[torch.Tensor]:
        """ppq/executor/torch.py:654-682: run only `operations` (already in execution order) on the
        given feeds -- the block forward of the training based passes.  Like ``forward`` it records no autograd
        graph unless ``with_gradient`` (the finetuning passes' training step) asks for one."""
        if not with_gradient:
            with torch.no_grad(): return self._partial_graph_forward(operations, feed_dict, output_names)
        return self._partial_graph_forward(operations, feed_dict, output_names)

    def _partial_graph_forward(self, operations, feed_dict, output_names):
        g = self._graph
        for name, value in feed_dict.items(): g.variables[name].value = self._place(value)
        results = [None] * len(output_names)
        self._fused_parameters(operations)            # the multi-tensor plan covers the parameters of THESE operations only
        for op in operations:
            raw_in = [v.value for v in op.inputs]
            if any(x is None for x in raw_in):
                raise ValueError(f'partial_graph_forward: input of {op.name} was not fed')
            if isinstance(op, QuantableOperation):
                qin = [self._quantize_parameter(v, c) if v.is_parameter else self.quantize_function(x, c)
                       for v, x, c in zip(op.inputs, raw_in, op.config.input_quantization_config)]
            else: qin = raw_in
            outs = _forward(op, qin)
            outs = list(outs) if isinstance(outs, (list, tuple)) else [outs]
            if isinstance(op, QuantableOperation):
                outs = [self.quantize_function(y, c) for y, c in zip(outs, op.config.output_quantization_config)]
            for v, y in zip(op.outputs, outs):
                v.value = y
                if v.name in output_names: results[output_names.index(v.name)] = y
        for v in g.variables.values():
            if not v.is_parameter: v.value = None
        return results

    def _fused_parameters(self, operations=None) -> None:
        """Fake-quantise every parameter whose config is an activated, non-delegated LINEAR one with a
        single multi-tensor launch (ffi.LinearQuantizePlan -> ppqhip_fq_linear_multi) instead of one
        launch per weight; same per-forward work as the reference executor (torch.py:516-518), same
        values as PPQLinearQuantFunction.  The device job table holds POINTERS, so in-place updates of
        weights / scales need no rebuild; replaced tensors or edited configs do (signature check)."""
        self._fused = {}
        if not self.fuse_parameter_quantization or self._default_quant_fn is not PPQuantFunction: return
        from .ffi import FloatingQuantizePlan, LinearQuantizePlan
        todo, sig = [], []
        for op in (self._graph.operations.values() if operations is None else operations):
            if not isinstance(op, QuantableOperation): continue
            for v, c in zip(op.inputs, op.config.input_quantization_config):
                if not (v.is_parameter and isinstance(v.value, torch.Tensor) and v.value.is_cuda): continue
                if not QuantizationStates.is_activated(c.state) or c in self._delegates: continue
                pol = c.policy
                floating = pol.has_property(P.FLOATING)
                if not (pol.has_property(P.LINEAR) or floating) or pol.has_property(P.DYNAMIC): continue
                if v.value.dtype != torch.float32 or v.value.requires_grad or v.value.numel() == 0: continue
                if not (isinstance(c.scale, torch.Tensor) and isinstance(c.offset, torch.Tensor)): continue
                axis = c.channel_axis if pol.has_property(P.PER_CHANNEL) else None
                if not LinearQuantizePlan.accepts(v.value, c.scale, c.offset, axis): continue   # per-tensor launch instead
                rnd = int(getattr(c.rounding, 'value', c.rounding))
                todo.append((v, c, axis, rnd, floating))
                sig.append((v.name, id(c), v.value.data_ptr(), c.scale.data_ptr(), c.offset.data_ptr(), tuple(v.value.shape), c.scale.numel(),
                            axis, c.quant_min, c.quant_max, rnd, floating,
                            (c.exponent_bits, c.mantissa_bits) if floating else None))
        if not todo:
            self._plans, self._plan_signature = [], None
            return
        cached = self._plan_cache.get(tuple(sig))
        if cached is not None:
            self._plan_cache[tuple(sig)] = self._plan_cache.pop(tuple(sig))          # most recently used last
            self._plans, self._plan_signature = cached, sig
        if sig != self._plan_signature:
            groups: Dict[tuple, list] = {}
            for item in todo: groups.setdefault((item[3], item[4]), []).append(item)
            self._plans = []
            for (rnd, floating), items in groups.items():
                if floating:        # TRT_FP8 policy: per-channel FP8 weights (one launch for all of them too)
                    plan = FloatingQuantizePlan([(v.value, c.scale, c.offset, axis, c.exponent_bits, c.mantissa_bits,
                                                  c.quant_min, c.quant_max) for v, c, axis, _, _ in items], rounding=rnd)
                else:
                    plan = LinearQuantizePlan([(v.value, c.scale, c.offset, axis, c.quant_min, c.quant_max)
                                               for v, c, axis, _, _ in items], rounding=rnd)
                self._plans.append((plan, [(v.name, id(c)) for v, c, _, _, _ in items]))
            self._plan_signature = sig
            if len(self._plan_cache) >= 8: self._plan_cache.pop(next(iter(self._plan_cache)))   # block / whole-graph plans alternate
            self._plan_cache[tuple(sig)] = self._plans
        for plan, keys in self._plans:
            for key, out in zip(keys, plan.run()): self._fused[key] = out

    def _quantize_parameter(self, var: Variable, config) -> torch.Tensor:
        hit = self._fused.get((var.name, id(config)))
        if hit is not None: return hit
        """A parameter and its scale do not change between calibration forwards, so its fake-quantised
        value is computed once and kept resident (the reference recomputes it every forward until
        ParameterBakingPass; same values, ppq/executor/torch.py:516-518).  The cache entry is keyed on
        the identity + version of value, scale and offset and on the config state."""
        if not self.cache_parameter_quantization or not QuantizationStates.is_activated(config.state):
            return self.quantize_function(var.value, config)
        key = (var.value.data_ptr(), var.value._version, id(config.scale), config.scale._version,
               id(config.offset), config.offset._version, int(getattr(config.state, 'value', config.state)))
        hit = self._param_cache.get(var.name)
        if hit is not None and hit[0] == key: return hit[1]
        q = self.quantize_function(var.value, config)
        self._param_cache[var.name] = (key, q)
        return q

    @ torch.no_grad()
    def forward_cached(self, inputs, output_names: List[str], cache: Dict[str, torch.Tensor]) -> List[torch.Tensor]:
        """``forward(inputs, output_names)`` that REUSES the quantised activations in ``cache`` (variable name -> tensor) and
        adds every tensor it computes: only the operations between what is cached and what is asked for run.  The values are
        those of ``forward`` as long as the owner drops the entries that depend on parameters / scales it changed
        (blocks.PrefixCache does).  A block-wise pass asks for the inputs of block after block: with the cache the quantised
        prefix of the graph is walked once per batch overall instead of once per block."""
        g = self._graph
        if isinstance(inputs, torch.Tensor): inputs = {next(iter(g.inputs)): inputs}
        elif isinstance(inputs, (list, tuple)): inputs = {k: v for k, v in zip(g.inputs, inputs)}
        for name, value in inputs.items():
            if name not in cache: cache[name] = self._place(value)
        need, stack = set(), [g.variables[n] for n in output_names]
        while stack:
            v = stack.pop()
            if v.name in cache or v.is_parameter: continue
            op = v.source_op
            if op is None: raise ValueError(f'forward_cached: graph input {v.name} was not fed')
            if op.name in need: continue
            need.add(op.name)
            stack.extend(op.inputs)
        ops = [op for op in g.topological_sort() if op.name in need]
        if ops: self._fused_parameters(ops)
        for op in ops:
            raw_in = [v.value if v.is_parameter else cache[v.name] for v in op.inputs]
            if isinstance(op, QuantableOperation):
                qin = [self._quantize_parameter(v, c) if v.is_parameter else self.quantize_function(x, c)
                       for v, x, c in zip(op.inputs, raw_in, op.config.input_quantization_config)]
            else: qin = raw_in
            outs = _forward(op, qin)
            outs = list(outs) if isinstance(outs, (list, tuple)) else [outs]
            if isinstance(op, QuantableOperation):
                outs = [self.quantize_function(y, c) for y, c in zip(outs, op.config.output_quantization_config)]
            for v, y in zip(op.outputs, outs): cache[v.name] = y
        return [cache[n] for n in output_names]

    @ torch.no_grad()
    def forward(self, inputs, output_names: List[str] = None, hooks: Dict[str, object] = None) -> List[torch.Tensor]:
        g = self._graph
        if isinstance(inputs, torch.Tensor): inputs = {next(iter(g.inputs)): inputs}
        elif isinstance(inputs, (list, tuple)): inputs = {k: v for k, v in zip(g.inputs, inputs)}
        for name, value in inputs.items(): g.variables[name].value = self._place(value)
        # explicit output names and no hooks: the walk ends with the last requested tensor (a block's quantised inputs need
        # the graph only up to the block; the reference walks on to the end, torch.py:499-570 -- same values)
        stop_early = output_names is not None and not hooks
        if output_names is None: output_names = list(g.outputs)
        results = [None] * len(output_names)
        visited = set()
        self._fused_parameters()
        for op in g.topological_sort():
            hook = hooks.get(op.name) if hooks else None
            raw_in = [v.value for v in op.inputs]
            qin = raw_in
            quantable = isinstance(op, QuantableOperation)
            if quantable:
                in_cfgs = list(op.config.input_quantization_config)
                qin = [self._quantize_parameter(v, c) if v.is_parameter else self.quantize_function(x, c)
                       for v, x, c in zip(op.inputs, raw_in, in_cfgs)]
            if hook is not None:
                qin = hook.pre_forward_hook(inputs=raw_in, quant_inputs=qin, quant_configs=in_cfgs)
            outs = _forward(op, qin)
            outs = list(outs) if isinstance(outs, (list, tuple)) else [outs]
            fp_outs = outs
            if quantable:
                out_cfgs = list(op.config.output_quantization_config)
                outs = [self.quantize_function(y, c) for y, c in zip(outs, out_cfgs)]
            if hook is not None:
                outs = hook.post_forward_hook(outputs=fp_outs, quant_outputs=outs, quant_configs=out_cfgs)
            for v, y in zip(op.outputs, outs):
                v.value = y
                if v.name in output_names: results[output_names.index(v.name)] = y
            visited.add(op.name)
            for v in op.inputs:                      # runtime clear, torch.py:564-568
                if not v.is_parameter and all(d.name in visited for d in v.dest_ops): v.value = None
            if stop_early and all(r is not None for r in results): break      # nothing downstream was asked for
        for v in g.variables.values():
            if not v.is_parameter: v.value = None
        return results


# ------------------------------------------------------------------------------------ quantizer
def quantize_graph(graph: BaseGraph, activation_algorithm: str = 'kl', per_channel_weight: bool = True,
                   symmetrical: bool = True, weight_symmetrical: bool = True, num_of_bits: int = 8,
                   hist_bins: int = None, fp8: bool = False, passive_bias: bool = False) -> None:
    """TensorRT-style INT8 policy (TensorRTQuantizer.py:12-107): per-tensor activations on every
    operation, per-channel `minmax` weights on axis 0, FP32 bias; then the state edits of
    QuantizeFusionPass (computing op -> activation, passive ops) and QuantizeSimplifyPass."""
    qmin, qmax = (-(2 ** (num_of_bits - 1)), 2 ** (num_of_bits - 1) - 1) if symmetrical else (0, 2 ** num_of_bits - 1)
    wmin, wmax = (-(2 ** (num_of_bits - 1)), 2 ** (num_of_bits - 1) - 1) if weight_symmetrical else (0, 2 ** num_of_bits - 1)

    def act_cfg():
        if fp8:
            from .core import FloatingQuantizationConfig
            return FloatingQuantizationConfig(calibration='floating')
        c = LinearQuantizationConfig(symmetrical=symmetrical, quant_min=qmin, quant_max=qmax, num_of_bits=num_of_bits,
                                     calibration=activation_algorithm)
        if hist_bins is not None: c.detail['OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE'] = hist_bins
        return c

    for name, op in list(graph.operations.items()):
        in_cfgs = []
        for i, v in enumerate(op.inputs):
            if v.is_parameter and op.type in COMPUTING_OP and i == 1:
                in_cfgs.append(LinearQuantizationConfig(symmetrical=weight_symmetrical, quant_min=wmin, quant_max=wmax,
                                                        num_of_bits=num_of_bits, calibration='minmax',
                                                        channel_axis=0 if per_channel_weight else None))
            elif v.is_parameter and passive_bias and not fp8 and op.type in COMPUTING_OP and i == 2:
                # the integer platforms' policy (PPLQuantizer.py:54-66): a 32-bit symmetric bias whose scale is NOT observed but
                # derived -- input scale x weight scale -- by PassiveParameterQuantizePass once both are calibrated
                in_cfgs.append(LinearQuantizationConfig(symmetrical=True, quant_min=-(2 ** 31 - 1), quant_max=2 ** 31 - 1, num_of_bits=32,
                                                        calibration=None, channel_axis=0 if per_channel_weight else None))
                in_cfgs[-1].state = QuantizationStates.PASSIVE_INIT
            elif v.is_parameter:
                c = act_cfg(); c.state = QuantizationStates.FP32      # bias / norm parameters stay FP32
                in_cfgs.append(c)
            else:
                in_cfgs.append(act_cfg())
        qop = QuantableOperation(op, OperationQuantizationConfig(in_cfgs, [act_cfg() for _ in op.outputs]))
        for v in qop.inputs: v.dest_ops[v.dest_ops.index(op)] = qop
        for v in qop.outputs: v.source_op = qop
        graph.operations[name] = qop
    ops = list(graph.operations.values())
    # The states AND the dominance links the reference's refine passes leave (TensorQuantizationConfig.dominated_by: an
    # OVERLAPPED config reads its root's scale / offset, which is what PassiveParameterQuantizePass multiplies):
    # QuantizeSimplifyPass (optim/refine.py): an input produced by a quantable op is already quantised by its producer
    for op in ops:
        for v, c in zip(op.inputs, op.config.input_quantization_config):
            if v.source_op is not None and is_initial(c):
                src = v.source_op
                c.dominated_by = src.config.output_quantization_config[src.outputs.index(v)]        # -> OVERLAPPED
    # QuantizeFusionPass: computing op followed by a single activation -> the conv output is not quantised on its own (the
    # activation's output config rules it); passive operations share their input's quantisation
    for op in ops:
        out = op.outputs[0]
        if op.type in COMPUTING_OP | {'Add'} and len(out.dest_ops) == 1 and out.dest_ops[0].type in {'Relu', 'Gelu'}:
            op.config.output_quantization_config[0].dominated_by = out.dest_ops[0].config.output_quantization_config[0]
        if op.type in PASSIVE_OPERATIONS:
            first = op.config.input_quantization_config[0]
            if first.dominated_by is not op.config.output_quantization_config[0].dominated_by:
                op.config.output_quantization_config[0].dominated_by = first
            else: op.config.output_quantization_config[0].state = QuantizationStates.OVERLAPPED


from .parameters import ParameterBakingPass, ParameterQuantizePass  # noqa: E402,F401  (kept under their old names here)


# ------------------------------------------------------------------------------------ topologies
def _he(shape, gen) -> torch.Tensor:
    fan_in = shape[1] * (shape[2] * shape[3] if len(shape) == 4 else 1)
    return torch.randn(shape, generator=gen) * (2.0 / fan_in) ** 0.5


def resnet50_graph(seed: int = 0, num_classes: int = 1000) -> BaseGraph:
    """ResNet-50 v1.5 topology, BatchNorm folded into the convolutions (PPQ's FORMATTER_FUSE_BN,
    ppq/core/common.py:40), seeded He-initialised weights, small random biases."""
    gen = torch.Generator().manual_seed(seed)
    g = BaseGraph('resnet50')
    x = g.create_variable('input')
    g.inputs['input'] = x
    n = [0]

    def conv(inp, cin, cout, k, stride=1, pad=0, relu=True, tag='conv'):
        n[0] += 1
        w = g.create_variable(f'{tag}{n[0]}_w', _he([cout, cin, k, k], gen), True)
        b = g.create_variable(f'{tag}{n[0]}_b', torch.randn(cout, generator=gen) * 0.05, True)
        y = g.create_operation('Conv', f'{tag}{n[0]}', [inp, w, b], {'strides': stride, 'pads': pad})
        if relu: y = g.create_operation('Relu', f'relu{n[0]}', [y])
        return y

    y = conv(x, 3, 64, 7, 2, 3)
    y = g.create_operation('MaxPool', 'maxpool', [y], {'kernel_shape': 3, 'strides': 2, 'pads': 1})
    cin = 64
    for stage, (blocks, width) in enumerate(zip([3, 4, 6, 3], [64, 128, 256, 512])):
        for blk in range(blocks):
            stride = 2 if (blk == 0 and stage > 0) else 1
            identity = y
            z = conv(y, cin, width, 1)
            z = conv(z, width, width, 3, stride, 1)
            z = conv(z, width, width * 4, 1, relu=False)
            if blk == 0:
                identity = conv(y, cin, width * 4, 1, stride, 0, relu=False, tag='down')
            n[0] += 1
            z = g.create_operation('Add', f'add{n[0]}', [z, identity])
            y = g.create_operation('Relu', f'relu{n[0]}', [z])
            cin = width * 4
    y = g.create_operation('GlobalAveragePool', 'gap', [y])
    y = g.create_operation('Flatten', 'flatten', [y])
    w = g.create_variable('fc_w', torch.randn([num_classes, 2048], generator=gen) * (1.0 / 2048) ** 0.5, True)
    b = g.create_variable('fc_b', torch.zeros(num_classes), True)
    y = g.create_operation('Gemm', 'fc', [y, w, b])
    g.outputs[y.name] = y
    return g


def yolov6s_graph(seed: int = 0, num_classes: int = 80) -> BaseGraph:
    """A YOLOv6-s-like detector (BASELINE config 5), deploy form: EfficientRep backbone (RepVGG blocks
    re-parameterised to 3x3 Conv + Relu, widths 32-64-128-256-512, repeats 1-2-4-6-2, SPPF-style pooling
    tail), Rep-PAN neck (1x1 reduce, nearest x2 Resize, Concat, Rep blocks; 3x3 stride-2 down path) and a
    decoupled head per scale (1x1 stem, 3x3 + 1x1 class branch, 3x3 + 1x1 box branch): 6 outputs at strides
    8 / 16 / 32.  Seeded He-initialised weights, BatchNorm folded.  Stands in for the ONNX model that cannot be
    loaded here (no `onnx`); what matters to this package is the operator mix the finetuning passes walk:
    plain conv chains, fan-outs that close at a Concat, Resize / MaxPool passive ops, multiple outputs."""
    gen = torch.Generator().manual_seed(seed)
    g = BaseGraph('yolov6s')
    x = g.create_variable('input')
    g.inputs['input'] = x
    n = [0]

    def conv(inp, cin, cout, k=3, stride=1, relu=True, tag='conv'):
        n[0] += 1
        # He-UNIFORM weights: every output channel spreads over its whole per-channel range, so INT4 (16 levels)
        # keeps information in nearly every weight -- like a trained, quantisation-aware detector, and unlike a
        # raw re-parameterised RepVGG kernel whose identity tap would own the range and round the rest to zero
        a = (6.0 / (cin * k * k)) ** 0.5
        w = g.create_variable(f'{tag}{n[0]}_w', (torch.rand([cout, cin, k, k], generator=gen) * 2 - 1) * a, True)
        b = g.create_variable(f'{tag}{n[0]}_b', torch.randn(cout, generator=gen) * 0.05, True)
        y = g.create_operation('Conv', f'{tag}{n[0]}', [inp, w, b], {'strides': stride, 'pads': k // 2})
        if relu: y = g.create_operation('Relu', f'relu{n[0]}', [y])
        return y

    def rep(inp, c, repeats):
        for _ in range(repeats): inp = conv(inp, c, c, 3, tag='rep')
        return inp

    def cat(name, vs): return g.create_operation('Concat', name, vs, {'axis': 1})

    y = conv(x, 3, 32, 3, 2, tag='stem')                         # stride 2
    feats, cin = [], 32
    for stage, (c, r) in enumerate(zip([64, 128, 256, 512], [2, 4, 6, 2])):
        y = conv(y, cin, c, 3, 2, tag=f's{stage}_down')
        y = rep(y, c, r)
        cin = c
        feats.append(y)                                          # strides 4, 8, 16, 32
    # SPPF-style tail on the stride-32 feature
    p = conv(feats[3], 512, 256, 1, tag='sppf_in')
    m1 = g.create_operation('MaxPool', 'sppf_pool1', [p], {'kernel_shape': 5, 'strides': 1, 'pads': 2})
    m2 = g.create_operation('MaxPool', 'sppf_pool2', [m1], {'kernel_shape': 5, 'strides': 1, 'pads': 2})
    m3 = g.create_operation('MaxPool', 'sppf_pool3', [m2], {'kernel_shape': 5, 'strides': 1, 'pads': 2})
    c5 = conv(cat('sppf_cat', [p, m1, m2, m3]), 1024, 512, 1, tag='sppf_out')
    c3, c4 = feats[1], feats[2]                                  # 128 @ s8, 256 @ s16
    # Rep-PAN: top-down
    r5 = conv(c5, 512, 128, 1, tag='reduce5')
    u5 = g.create_operation('Resize', 'up5', [r5], {'scale': 2})
    p4 = rep(conv(cat('cat_p4', [u5, c4]), 128 + 256, 128, 3, tag='p4_in'), 128, 3)
    r4 = conv(p4, 128, 64, 1, tag='reduce4')
    u4 = g.create_operation('Resize', 'up4', [r4], {'scale': 2})
    p3 = rep(conv(cat('cat_p3', [u4, c3]), 64 + 128, 64, 3, tag='p3_in'), 64, 3)          # out @ s8
    # bottom-up
    d3 = conv(p3, 64, 64, 3, 2, tag='down3')
    n4 = rep(conv(cat('cat_n4', [d3, r4]), 64 + 64, 128, 3, tag='n4_in'), 128, 3)         # out @ s16
    d4 = conv(n4, 128, 128, 3, 2, tag='down4')
    n5 = rep(conv(cat('cat_n5', [d4, r5]), 128 + 128, 256, 3, tag='n5_in'), 256, 3)       # out @ s32
    for name, f, c in (('s8', p3, 64), ('s16', n4, 128), ('s32', n5, 256)):
        stem = conv(f, c, c, 1, tag=f'head_{name}_stem')
        cls = conv(conv(stem, c, c, 3, tag=f'head_{name}_cls'), c, num_classes, 1, relu=False, tag=f'head_{name}_cls_out')
        box = conv(conv(stem, c, c, 3, tag=f'head_{name}_box'), c, 4, 1, relu=False, tag=f'head_{name}_box_out')
        g.outputs[cls.name] = cls
        g.outputs[box.name] = box
    _unit_variance_init(g, torch.rand(2, 3, 96, 96, generator=gen))
    return g


@ torch.no_grad()
def _unit_variance_init(g: BaseGraph, sample: torch.Tensor) -> None:
    """Layer-sequential unit-variance rescaling of the seeded weights (a trained, BatchNorm-folded detector has
    O(1) activations everywhere; 56 unnormalised He-initialised convolutions in a row do not: the positive mean
    of ReLU outputs compounds to 1e10).  One CPU forward on a seeded sample; every Conv's weight and bias are
    divided by the standard deviation of its output.  Deterministic in `seed`."""
    values = {next(iter(g.inputs)): sample}
    for op in g.operations.values():
        xs = [v.value if v.is_parameter else values[v.name] for v in op.inputs]
        y = _forward(op, xs)
        if op.type == 'Conv':
            std = float(y.std())
            if std > 0:
                op.inputs[1].value = op.inputs[1].value / std
                if len(op.inputs) > 2: op.inputs[2].value = op.inputs[2].value / std
                y = y / std
        values[op.outputs[0].name] = y


def small_cnn_graph(seed: int = 0, width: int = 16) -> BaseGraph:
    """Conv-Relu-Conv-Add-Relu-GAP-Gemm: a tiny graph for smoke / parity tests."""
    gen = torch.Generator().manual_seed(seed)
    g = BaseGraph('small_cnn')
    x = g.create_variable('input'); g.inputs['input'] = x

    def conv(inp, cin, cout, name, relu):
        w = g.create_variable(name + '_w', _he([cout, cin, 3, 3], gen), True)
        b = g.create_variable(name + '_b', torch.randn(cout, generator=gen) * 0.1, True)
        y = g.create_operation('Conv', name, [inp, w, b], {'strides': 1, 'pads': 1})
        return g.create_operation('Relu', name + '_relu', [y]) if relu else y

    a = conv(x, 3, width, 'c1', True)
    b_ = conv(a, width, width, 'c2', False)
    s = g.create_operation('Add', 'add', [b_, a])
    s = g.create_operation('Relu', 'add_relu', [s])
    p = g.create_operation('GlobalAveragePool', 'gap', [s])
    f = g.create_operation('Flatten', 'flatten', [p])
    w = g.create_variable('fc_w', torch.randn([10, width], generator=gen) * 0.3, True)
    bb = g.create_variable('fc_b', torch.zeros(10), True)
    y = g.create_operation('Gemm', 'fc', [f, w, bb])
    g.outputs[y.name] = y
    return g


def transformer_mlp_graph(seed: int = 0, dim: int = 64, hidden: int = 256) -> BaseGraph:
    """LayerNorm -> Gemm -> Gelu -> Gemm -> Add(residual): the MLP half of a ViT block, the operator
    mix BASELINE config 4 (ViT-B/16, FP8 E4M3) exercises."""
    gen = torch.Generator().manual_seed(seed)
    g = BaseGraph('transformer_mlp')
    x = g.create_variable('input'); g.inputs['input'] = x
    gamma = g.create_variable('ln_w', torch.ones(dim) + torch.randn(dim, generator=gen) * 0.05, True)
    beta = g.create_variable('ln_b', torch.randn(dim, generator=gen) * 0.05, True)
    h = g.create_operation('LayerNormalization', 'ln', [x, gamma, beta])
    w1 = g.create_variable('fc1_w', torch.randn([hidden, dim], generator=gen) * (1.0 / dim) ** 0.5, True)
    b1 = g.create_variable('fc1_b', torch.zeros(hidden), True)
    h = g.create_operation('Gemm', 'fc1', [h, w1, b1])
    h = g.create_operation('Gelu', 'gelu', [h])
    w2 = g.create_variable('fc2_w', torch.randn([dim, hidden], generator=gen) * (1.0 / hidden) ** 0.5, True)
    b2 = g.create_variable('fc2_b', torch.zeros(dim), True)
    h = g.create_operation('Gemm', 'fc2', [h, w2, b2])
    y = g.create_operation('Add', 'residual', [h, x])
    g.outputs[y.name] = y
    return g


def vit_graph(seed: int = 0, depth: int = 12, dim: int = 768, heads: int = 12, mlp_dim: int = 3072, patch: int = 16,
              image: int = 224, num_classes: int = 1000) -> BaseGraph:
    """ViT-B/16 topology (BASELINE config 4): patch-embedding Conv, class token + position embedding,
    `depth` pre-norm blocks (LayerNorm -> QKV Gemm -> scaled-dot-product attention with two MatMul ->
    projection Gemm -> residual; LayerNorm -> Gemm -> Gelu -> Gemm -> residual), final LayerNorm, head.
    Seeded random weights (no onnx / checkpoints in the image)."""
    gen = torch.Generator().manual_seed(seed)
    g = BaseGraph('vit')
    tokens, hd = (image // patch) ** 2 + 1, dim // heads

    def param(name, *shape, std=None, ones=False):
        if ones: val = torch.ones(*shape)
        elif std == 0: val = torch.zeros(*shape)
        else: val = torch.randn(*shape, generator=gen) * (std if std is not None else (1.0 / shape[-1]) ** 0.5)
        return g.create_variable(name, val, True)

    def gemm(inp, name, cin, cout):
        return g.create_operation('Gemm', name, [inp, param(name + '_w', cout, cin), param(name + '_b', cout, std=0)])

    def layer_norm(inp, name):
        return g.create_operation('LayerNormalization', name, [inp, param(name + '_w', dim, ones=True), param(name + '_b', dim, std=0)])
    x = g.create_variable('input'); g.inputs['input'] = x
    h = g.create_operation('Conv', 'patch_embed', [x, param('patch_w', dim, 3, patch, patch, std=(1.0 / (3 * patch * patch)) ** 0.5),
                                                  param('patch_b', dim, std=0)], {'strides': patch})
    h = g.create_operation('Reshape', 'patch_flat', [h], {'shape': (-1, dim, tokens - 1)})
    h = g.create_operation('Transpose', 'patch_tokens', [h], {'perm': (0, 2, 1)})
    h = g.create_operation('Concat', 'cat_cls', [param('cls_token', 1, 1, dim, std=0.02), h], {'axis': 1})
    h = g.create_operation('Add', 'add_pos', [h, param('pos_embed', 1, tokens, dim, std=0.02)])
    for i in range(depth):
        p = f'blk{i}_'
        a = layer_norm(h, p + 'ln1')
        qkv = gemm(a, p + 'qkv', dim, 3 * dim)
        parts = []
        for k, nm in enumerate('qkv'):
            t = g.create_operation('Slice', p + nm + '_slice', [qkv], {'axis': 2, 'start': k * dim, 'length': dim})
            t = g.create_operation('Reshape', p + nm + '_heads', [t], {'shape': (-1, tokens, heads, hd)})
            parts.append(g.create_operation('Transpose', p + nm + '_t', [t], {'perm': (0, 2, 3, 1) if nm == 'k' else (0, 2, 1, 3)}))
        att = g.create_operation('MatMul', p + 'qk', [parts[0], parts[1]])
        att = g.create_operation('Mul', p + 'scale', [att], {'value': hd ** -0.5})
        att = g.create_operation('Softmax', p + 'softmax', [att], {'axis': -1})
        ctx = g.create_operation('MatMul', p + 'av', [att, parts[2]])
        ctx = g.create_operation('Transpose', p + 'ctx_t', [ctx], {'perm': (0, 2, 1, 3)})
        ctx = g.create_operation('Reshape', p + 'ctx', [ctx], {'shape': (-1, tokens, dim)})
        h = g.create_operation('Add', p + 'res1', [gemm(ctx, p + 'proj', dim, dim), h])
        m = layer_norm(h, p + 'ln2')
        m = g.create_operation('Gelu', p + 'gelu', [gemm(m, p + 'fc1', dim, mlp_dim)])
        h = g.create_operation('Add', p + 'res2', [gemm(m, p + 'fc2', mlp_dim, dim), h])
    h = layer_norm(h, 'ln_f')
    h = g.create_operation('Slice', 'cls_out', [h], {'axis': 1, 'start': 0, 'length': 1})
    h = g.create_operation('Reshape', 'cls_flat', [h], {'shape': (-1, dim)})
    y = gemm(h, 'head', dim, num_classes)
    g.outputs[y.name] = y
    return g


def quantize_graph_fp8(graph: BaseGraph, exponent: int = 4, mantissa: int = 3, operations=None) -> None:
    """TRT_FP8-style policy (quantizer/FP8Quantizer.py:107-200): only the inputs of Conv / Gemm / MatMul
    are quantised -- activations per tensor with the power-of-2 'floating' observer, weights per
    channel (axis 0) with the same observer; everything else stays FP32.  `operations`: names of the
    operations a dispatcher chose to quantise (default: all; the reference's 'conservative' dispatcher
    leaves everything downstream of a non-quantable type such as Add on the FP32 platform)."""
    from .core import FloatingQuantizationConfig
    qmax = 448.0 if (exponent, mantissa) == (4, 3) else 57344.0
    for name, op in list(graph.operations.items()):
        if operations is not None and name not in operations: continue
        in_cfgs = []
        for i, v in enumerate(op.inputs):
            c = FloatingQuantizationConfig(exponent=exponent, mantissa=mantissa, quant_min=-qmax, quant_max=qmax,
                                           calibration='floating',
                                           channel_axis=0 if (v.is_parameter and i == 1) else None)
            if op.type not in COMPUTING_OP or (v.is_parameter and i != 1): c.state = QuantizationStates.FP32
            in_cfgs.append(c)
        outs = []
        for _ in op.outputs:
            c = FloatingQuantizationConfig(exponent=exponent, mantissa=mantissa, quant_min=-qmax, quant_max=qmax)
            c.state = QuantizationStates.FP32
            outs.append(c)
        qop = QuantableOperation(op, OperationQuantizationConfig(in_cfgs, outs))
        for v in qop.inputs: v.dest_ops[v.dest_ops.index(op)] = qop
        for v in qop.outputs: v.source_op = qop
        graph.operations[name] = qop
