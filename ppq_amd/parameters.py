"""Parameter-side passes next to the calibration pass (immediate neighbours of RuntimeCalibrationPass; 8f-3).

``ParameterQuantizePass`` mirrors ppq/quantization/optim/parameters.py:156-215: every INITIAL parameter config is observed and
rendered.  The reference reaches the weights with a dummy forward through per-operation hooks (one observer call, i.e. two
reductions, per weight); here the observers of ALL parameters ride one ``ObservationQueue`` -- per-channel ranges are ONE
``ppqhip_minmax_c_multi`` launch for the whole graph -- and one batched render.  Same scales (tests/test_gpu_reference.py).

``ParameterBakingPass`` mirrors ppq/quantization/optim/baking.py:11-47 (+ ``QuantableOperation.baking_parameters``,
IR/quantize.py:98-111): every ACTIVATED / PASSIVE parameter is replaced by its fake-quantised value and its config becomes
BAKED / PASSIVE_BAKED.  The LINEAR ones are fake-quantised by ONE ``ppqhip_fq_linear_multi`` launch per rounding policy.

``PassiveParameterQuantizePass`` mirrors ppq/quantization/optim/parameters.py:13-153: parameters that may not own a scale --
the bias of Conv / ConvTranspose / Gemm (scale = input scale x weight scale, offset 0), the bounds of Clip and the pad value of
Pad (mastered by the operation's input config).  Pure host logic: it only wires configs; the tensors keep living on the device
and the products are device tensor ops.

All three work on this package's harness graph and, duck-typed, on the reference's own ``BaseGraph`` (``ppq.lib.Pipeline``
accepts them after ``install_plugins_into_ppq()``).
"""
import torch

from .calibration import QuantizationOptimizationPass
from .core import QuantizationProperty as P
from .core import QuantizationStates, QuantizationVisibility, is_initial, rounding_value, state_value

_S = QuantizationStates
_QUANTIZED_INPUT = {_S.PASSIVE.value, _S.ACTIVATED.value, _S.BAKED.value, _S.OVERLAPPED.value}
_BIAS_OWNERS = {'Conv', 'ConvTranspose', 'Gemm'}


def _state_of(cfg, wanted):
    """The member named like ``wanted`` of the enum class ``cfg.state`` comes from (ours, or the reference's)."""
    cls = type(cfg.state)
    return getattr(cls, wanted.name) if hasattr(cls, wanted.name) else wanted


def _visibility_of(cfg, wanted):
    """The member named like ``wanted`` of the enum class ``cfg.visibility`` comes from: a config of the reference's keeps
    members of the reference's enum, so that its own ``can_export`` (an ``==`` against ITS ``INTERNAL``) still recognises them."""
    cls = type(getattr(cfg, 'visibility', wanted))
    return getattr(cls, wanted.name) if hasattr(cls, wanted.name) else wanted


class ParameterQuantizePass(QuantizationOptimizationPass):
    """optim/parameters.py:156-215.  ``method`` overrides the observer algorithm of every parameter config (:186-188);
    ``dataloader`` / ``executor`` are accepted for the reference's call protocol and not needed: the parameters are read
    where they lie (the reference's dummy forward exists only to reach them through its hooks)."""
    def __init__(self, method: str = None):
        super().__init__(name='PPQ Parameter Quantization Pass')
        self._method = method
        self.launches = 0                      # statistics launches of the last optimize() (all per-channel weights: ONE)

    def optimize(self, graph, dataloader=None, executor=None, **kwargs) -> None:
        from .observer import ObservationQueue, TensorObserverFactroy, render_observers
        observers, queue = [], ObservationQueue()
        for op in graph.operations.values():
            if not hasattr(op, 'config'): continue
            for config, var in op.config_with_variable:
                if not var.is_parameter: continue
                if self._method is not None: config.observer_algorithm = self._method
                if not is_initial(config): continue
                ob = TensorObserverFactroy.build_observer(var, config)
                ob.queue = queue                       # the statistics of ALL parameters: one launch per kind (per-channel
                ob.observe(var.value)                  # ranges -> ppqhip_minmax_c_multi), flushed by the first render
                observers.append(ob)
        queue.flush()
        self.launches = queue.launches
        render_observers(observers)
        for ob in observers: ob.queue = None


class ParameterBakingPass(QuantizationOptimizationPass):
    """optim/baking.py:11-47 + IR/quantize.py:98-111.  ``fused``: the non-dynamic LINEAR (and FP8) configs whose tensors the
    multi-tensor plans can point at are fake-quantised together (``ffi.LinearQuantizePlan`` / ``FloatingQuantizePlan``, one
    launch per rounding policy; values identical to the per-tensor function, tests/test_gpu_calibration.py); everything else --
    dynamic, a custom ``quantize_function``, an oddly laid out tensor -- goes through ``quantize_function`` one by one, as in
    the reference."""
    def __init__(self, quantize_function=None, fused: bool = True) -> None:
        super().__init__(name='PPQ Parameter Baking Pass')
        if quantize_function is None:
            from .qfunction import PPQuantFunction
            quantize_function = PPQuantFunction
        self._quantize_function = quantize_function
        self._fused = fused
        self.launches = 0                      # multi-tensor launches of the last optimize()
        self.per_tensor = 0                    # parameters that went through quantize_function instead

    @ staticmethod
    def _baked_state(config):
        cls = type(config.state)
        name = 'BAKED' if state_value(config.state) == _S.ACTIVATED.value else 'PASSIVE_BAKED'
        return getattr(cls, name) if hasattr(cls, name) else getattr(_S, name)

    def optimize(self, graph, **kwargs) -> None:
        from .ffi import FloatingQuantizePlan, LinearQuantizePlan
        from .qfunction import PPQuantFunction
        todo = []
        for op in graph.operations.values():
            if not hasattr(op, 'config'): continue
            for config, var in op.config_with_variable:
                if var.is_parameter and state_value(config.state) in (_S.ACTIVATED.value, _S.PASSIVE.value):
                    assert len(var.dest_ops) == 1, (
                        f', Parameter {var.name} has {len(var.dest_ops)} destinations, Baking parameter that has more than 1 '
                        'destinations will incur unexpected problems, PPQ does not support parameters with more than 1 related '
                        'operation, reform your graph first.')
                    todo.append((config, var))
        groups, single = {}, []
        for config, var in todo:
            pol, v = config.policy, var.value
            axis = config.channel_axis if pol.has_property(P.PER_CHANNEL) else None
            floating = pol.has_property(P.FLOATING)
            fusable = (self._fused and self._quantize_function is PPQuantFunction and isinstance(v, torch.Tensor) and v.is_cuda
                       and v.dtype == torch.float32 and v.numel() > 0 and not v.requires_grad
                       and (pol.has_property(P.LINEAR) or floating) and not pol.has_property(P.DYNAMIC)
                       and isinstance(config.scale, torch.Tensor) and isinstance(config.offset, torch.Tensor)
                       and config.scale.dtype == torch.float32 and config.offset.dtype == torch.float32
                       and config.scale.device == v.device and config.offset.device == v.device
                       and LinearQuantizePlan.accepts(v, config.scale, config.offset, axis))
            if fusable: groups.setdefault((rounding_value(config.rounding), v.device, floating), []).append((config, var, axis))
            else: single.append((config, var))
        self.launches, self.per_tensor = 0, len(single)
        for (rnd, _, floating), items in groups.items():
            if len(items) == 1:
                single.append(items[0][:2]); self.per_tensor += 1
                continue
            if floating:                            # TRT_FP8 policy: the per-channel FP8 weights, ppqhip_fq_float_multi
                plan = FloatingQuantizePlan([(var.value, c.scale, c.offset, axis, c.exponent_bits, c.mantissa_bits, c.quant_min,
                                              c.quant_max) for c, var, axis in items], rounding=rnd)
            else:
                plan = LinearQuantizePlan([(var.value, c.scale, c.offset, axis, c.quant_min, c.quant_max) for c, var, axis in items],
                                          rounding=rnd)
            outs = plan.run()
            self.launches += 1
            for (c, var, _), out in zip(items, outs):
                var.value = out                     # a view of the plan's arena: the baked parameters of a graph share ONE allocation
                c.state = self._baked_state(c)
        for c, var in single:
            var.value = self._quantize_function(var.value, c)
            c.state = self._baked_state(c)


class PassiveParameterQuantizePass(QuantizationOptimizationPass):
    """optim/parameters.py:13-153.

    * ``process_bias``: a PASSIVE_INIT (or already PASSIVE) bias config of Conv / ConvTranspose / Gemm gets
      ``scale = weight_scale * input_scale``, ``offset = 0``, state PASSIVE; a multi-dimensional bias is squeezed to [C];
    * ``process_clip`` / ``process_pad``: the min / max of Clip and the pad value of a 3-input Pad are mastered by the
      operation's input config (they must be representable in the input's grid) and get the given visibility.
    The operation's input must already be quantised (PASSIVE / ACTIVATED / BAKED / OVERLAPPED), else PermissionError, as in
    the reference; asymmetric passive parameters are refused."""
    def __init__(self, process_clip: bool = True, process_bias: bool = True, process_pad: bool = True,
                 clip_visiblity=QuantizationVisibility.INTERNAL, pad_visiblity=QuantizationVisibility.INTERNAL):
        super().__init__(name='PPQ Passive Parameter Quantization')
        self.process_clip, self.process_bias, self.process_pad = process_clip, process_bias, process_pad
        self.clip_visiblity, self.pad_visiblity = clip_visiblity, pad_visiblity
        self.unresolved = []                 # (op name, variable name) still PASSIVE_INIT afterwards (the reference warns)

    @ staticmethod
    def _require_quantized_input(op, cfg, what: str) -> None:
        if state_value(cfg.state) not in _QUANTIZED_INPUT:
            raise PermissionError(f'Can not quantize {what} of layer {op.name}, cause input has not been correctly quantized.')

    def _bias(self, op) -> None:
        if len(op.inputs) != 3: return
        i_cfg, w_cfg, b_cfg = op.config.input_quantization_config
        if state_value(b_cfg.state) not in (_S.PASSIVE.value, _S.PASSIVE_INIT.value): return
        var = op.inputs[-1]
        bias = var.value
        if bias is None:
            raise ValueError(f'Bias Varaible {var.name} must be a constant. Please check it again.')
        assert bias.numel() == bias.shape[-1], (f'For op {op.name}, expect Bias shape to be {[bias.numel()]}, '
                                                f'however {bias.shape} was given')
        var.value = bias.reshape(-1) if bias.ndim != 1 else bias          # [1, .., C] -> [C]; a single number stays [1]
        self._require_quantized_input(op, i_cfg, 'bias')
        assert not b_cfg.policy.has_property(P.ASYMMETRICAL), 'Passive parameter does not support ASYMMETRICAL quantization'
        b_cfg.scale = w_cfg.scale * i_cfg.scale
        b_cfg.offset = torch.zeros_like(b_cfg.scale)
        b_cfg.state = _state_of(b_cfg, _S.PASSIVE)

    def optimize(self, graph, **kwargs) -> None:
        for op in graph.operations.values():
            if not hasattr(op, 'config'): continue
            cfgs = op.config.input_quantization_config
            if self.process_bias and op.type in _BIAS_OWNERS: self._bias(op)
            if self.process_clip and op.type == 'Clip':
                self._require_quantized_input(op, cfgs[0], 'clip value')
                for cfg in cfgs[1:]:
                    cfg.master_by = cfgs[0]
                    cfg.visibility = _visibility_of(cfg, self.clip_visiblity)
            if self.process_pad and op.type == 'Pad' and len(op.inputs) == 3:
                self._require_quantized_input(op, cfgs[0], 'pad value')
                if len(cfgs) > 1:
                    cfgs[-1].master_by = cfgs[0]
                    cfgs[-1].visibility = _visibility_of(cfgs[-1], self.pad_visiblity)
        self.unresolved = [(op.name, var.name) for op in graph.operations.values() if hasattr(op, 'config')
                           for cfg, var in op.config_with_variable if state_value(cfg.state) == _S.PASSIVE_INIT.value]
        for op_name, var_name in self.unresolved:
            print(f'[Warning] Unexpected quantization state of variable {var_name} at op {op_name}, The configuration state has '
                  'been initialized as PASSIVE_INIT, however PassiveParameterQuantizePass do not kown how to deal with it.')
