"""Parameter-side passes next to the calibration pass (immediate neighbours of RuntimeCalibrationPass).

``PassiveParameterQuantizePass`` mirrors ppq/quantization/optim/parameters.py:13-153: parameters that may not own a scale --
the bias of Conv / ConvTranspose / Gemm (scale = input scale x weight scale, offset 0), the bounds of Clip and the pad value of
Pad (mastered by the operation's input config).  Pure host logic: it only wires configs; the tensors keep living on the device
and the products are device tensor ops.  Works on this package's harness graph and, duck-typed, on the reference's own
``BaseGraph`` (``ppq.lib.Pipeline`` accepts it after ``install_plugins_into_ppq()``).
"""
import torch

from .calibration import QuantizationOptimizationPass
from .core import QuantizationProperty as P
from .core import QuantizationStates, QuantizationVisibility, state_value

_S = QuantizationStates
_QUANTIZED_INPUT = {_S.PASSIVE.value, _S.ACTIVATED.value, _S.BAKED.value, _S.OVERLAPPED.value}
_BIAS_OWNERS = {'Conv', 'ConvTranspose', 'Gemm'}


def _state_of(cfg, wanted):
    """The member named like ``wanted`` of the enum class ``cfg.state`` comes from (ours, or the reference's)."""
    cls = type(cfg.state)
    return getattr(cls, wanted.name) if hasattr(cls, wanted.name) else wanted


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
                    cfg.visibility = self.clip_visiblity
            if self.process_pad and op.type == 'Pad' and len(op.inputs) == 3:
                self._require_quantized_input(op, cfgs[0], 'pad value')
                if len(cfgs) > 1:
                    cfgs[-1].master_by = cfgs[0]
                    cfgs[-1].visibility = self.pad_visiblity
        self.unresolved = [(op.name, var.name) for op in graph.operations.values() if hasattr(op, 'config')
                           for cfg, var in op.config_with_variable if state_value(cfg.state) == _S.PASSIVE_INIT.value]
        for op_name, var_name in self.unresolved:
            print(f'[Warning] Unexpected quantization state of variable {var_name} at op {op_name}, The configuration state has '
                  'been initialized as PASSIVE_INIT, however PassiveParameterQuantizePass do not kown how to deal with it.')
