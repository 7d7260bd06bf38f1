"""Fake-quant functions: the innermost hot function of PPQ's executor, over the HIP kernels.

Mirror of ppq/quantization/qfunction/{__init__,linear,floating}.py -- same function names, same
dispatch on the TensorQuantizationConfig policy bits and state, same straight-through backward
(``return dy, None ...`` linear.py:48-50), same errors.  The difference: there is no PyTorch
arithmetic branch; every activated config runs ``CUDA.LinearQuantize_T/C`` or
``CUDA.FloatingQuantize_T/C`` (ppq_amd/ffi.py), and a CPU tensor raises.

``TorchExecutor.quantize_function`` (ppq/executor/torch.py:610-613) calls ``PPQuantFunction`` by
default, so assigning ``executor._default_quant_fn = ppq_amd.qfunction.PPQuantFunction`` -- or just
installing the extension with ``ppq_amd.install_into_ppq()`` and keeping PPQ's own function --
routes the executor through these kernels.
"""
import torch
from torch.autograd import Function

from .core import QuantizationProperty as P
from .core import QuantizationStates, rounding_value, state_value
from .ffi import CUDA


class BaseQuantFunction:
    """qfunction/base.py:6-12: the callable protocol ``fn(tensor, config) -> tensor`` executors and delegators follow."""
    def __call__(self, input_tensor, quantization_config, **kwargs):
        raise NotImplementedError('Implement this first.')


def _activated(config) -> bool:
    return state_value(config.state) in (QuantizationStates.ACTIVATED.value, QuantizationStates.PASSIVE.value)


class TensorwiseLinearQuantImpl(Function):
    """qfunction/linear.py:8-50."""
    @ staticmethod
    def forward(ctx, tensor, scales, offsets, quant_min: int, quant_max: int, rounding) -> torch.Tensor:
        scales, offsets = scales.to(tensor.device), offsets.to(tensor.device)
        return CUDA.LinearQuantize_T(tensor=tensor, scales=scales, offsets=offsets, minimum=quant_min,
                                     maximum=quant_max, rounding=rounding_value(rounding))

    @ staticmethod
    def backward(ctx, dy: torch.Tensor):
        return dy, None, None, None, None, None


class ChannelwiseLinearQuantImpl(Function):
    """qfunction/linear.py:53-96."""
    @ staticmethod
    def forward(ctx, tensor, scales, offsets, channel_axis: int, quant_min: int, quant_max: int,
                rounding) -> torch.Tensor:
        scales, offsets = scales.to(tensor.device), offsets.to(tensor.device)
        return CUDA.LinearQuantize_C(tensor=tensor, scales=scales, offsets=offsets, channel_axis=channel_axis,
                                     minimum=quant_min, maximum=quant_max, rounding=rounding_value(rounding))

    @ staticmethod
    def backward(ctx, dy: torch.Tensor):
        return dy, None, None, None, None, None, None


class TensorwiseFloatingQuantImpl(Function):
    """qfunction/floating.py:7-49."""
    @ staticmethod
    def forward(ctx, tensor, scales, offsets, exponet_bits: int, mantissa_bits: int, quant_min: float,
                quant_max: float, rounding) -> torch.Tensor:
        scales, offsets = scales.to(tensor.device), offsets.to(tensor.device)
        return CUDA.FloatingQuantize_T(tensor=tensor, scales=scales, offsets=offsets, exponent=exponet_bits,
                                       mantissa=mantissa_bits, minimum=quant_min, maximum=quant_max,
                                       rounding=rounding_value(rounding))

    @ staticmethod
    def backward(ctx, dy: torch.Tensor):
        return dy, None, None, None, None, None, None, None


class ChannelwiseFloatingQuantImpl(Function):
    """qfunction/floating.py:52-92."""
    @ staticmethod
    def forward(ctx, tensor, scales, offsets, channel_axis: int, exponet_bits: int, mantissa_bits: int,
                quant_min: float, quant_max: float, rounding) -> torch.Tensor:
        scales, offsets = scales.to(tensor.device), offsets.to(tensor.device)
        return CUDA.FloatingQuantize_C(tensor=tensor, scales=scales, offsets=offsets, channel_axis=channel_axis,
                                       exponent=exponet_bits, mantissa=mantissa_bits, minimum=quant_min,
                                       maximum=quant_max, rounding=rounding_value(rounding))

    @ staticmethod
    def backward(ctx, dy: torch.Tensor):
        return dy, None, None, None, None, None, None, None, None


def _as_1d(t: torch.Tensor) -> torch.Tensor:
    """Per-tensor scale/offset arrive 0-d (``torch.tensor([s]).squeeze(0)``, observer/range.py:115)."""
    return t.reshape(1) if t.ndim == 0 else t


def PPQLinearQuantFunction(tensor: torch.Tensor, config) -> torch.Tensor:
    """qfunction/linear.py:200-216."""
    if not _activated(config): return tensor
    if not config.policy.has_property(P.LINEAR):
        raise ValueError('Critical Quantization Error! Non-linear config detected.')
    if config.policy.has_property(P.DYNAMIC):
        raise ValueError('Unexpected Dynamic Flag in Quantization Policy. Use PPQDyamicQuantFunction Instead.')
    if config.policy.has_property(P.PER_CHANNEL):
        return ChannelwiseLinearQuantImpl.apply(tensor, config.scale, config.offset, config.channel_axis,
                                                config.quant_min, config.quant_max, config.rounding)
    elif config.policy.has_property(P.PER_TENSOR):
        return TensorwiseLinearQuantImpl.apply(tensor, _as_1d(config.scale), _as_1d(config.offset),
                                               config.quant_min, config.quant_max, config.rounding)


class TensorwiseDynamicLinearQuantImpl(Function):
    """qfunction/linear.py:99-127: min / max of THIS tensor -> scale / offset -> fake quant (straight-through backward).
    The range reduction is the min/max kernel; the scale / offset arithmetic is ``minmax_to_scale_offset`` on Python floats,
    as in the reference (``tensor.min().item()``)."""
    @ staticmethod
    def forward(ctx, tensor: torch.Tensor, config) -> torch.Tensor:
        from .observer import minmax_to_scale_offset
        dev = tensor.device
        mm = torch.tensor([float('inf'), float('-inf')], device=dev)
        CUDA.MinMax_T(tensor, mm)
        mn, mx = mm.tolist()
        s, o = minmax_to_scale_offset(mn, mx, config)
        return CUDA.LinearQuantize_T(tensor=tensor, scales=torch.tensor([s], dtype=torch.float32, device=dev),
                                     offsets=torch.tensor([o], dtype=torch.float32, device=dev), minimum=config.quant_min,
                                     maximum=config.quant_max, rounding=rounding_value(config.rounding))

    @ staticmethod
    def backward(ctx, dy: torch.Tensor):
        return dy, None


class ChannelwiseDynamicLinearQuantImpl(Function):
    """qfunction/linear.py:130-172: the same per channel (``MinMax_C`` + the reference's loop over the channels' Python-float
    ranges -- double arithmetic, NOT the float32 array arithmetic of the observers' per-channel render)."""
    @ staticmethod
    def forward(ctx, tensor: torch.Tensor, config) -> torch.Tensor:
        from .observer import minmax_to_scale_offset
        dev = tensor.device
        C = tensor.shape[config.channel_axis]
        mins = torch.full([C], float('inf'), device=dev); maxs = torch.full([C], float('-inf'), device=dev)
        CUDA.MinMax_C(tensor, config.channel_axis, mins, maxs)
        so = [minmax_to_scale_offset(a, b, config) for a, b in zip(mins.tolist(), maxs.tolist())]
        scales = torch.tensor([v[0] for v in so], dtype=torch.float32, device=dev)
        offsets = torch.tensor([v[1] for v in so], dtype=torch.float32, device=dev)
        return CUDA.LinearQuantize_C(tensor=tensor, scales=scales, offsets=offsets, channel_axis=config.channel_axis,
                                     minimum=config.quant_min, maximum=config.quant_max, rounding=rounding_value(config.rounding))

    @ staticmethod
    def backward(ctx, dy: torch.Tensor):
        return dy, None


def PPQDyamicLinearQuantFunction(tensor: torch.Tensor, config) -> torch.Tensor:
    """qfunction/linear.py:174-198."""
    if not _activated(config): return tensor
    if not config.policy.has_property(P.LINEAR):
        raise ValueError('Critical Quantization Error! Non-linear config detected.')
    if not config.policy.has_property(P.DYNAMIC):
        raise ValueError('Quantization Policy Do Not Have Dynamic Flag!')
    if config.policy.has_property(P.PER_CHANNEL):
        return ChannelwiseDynamicLinearQuantImpl.apply(tensor, config)
    elif config.policy.has_property(P.PER_TENSOR):
        return TensorwiseDynamicLinearQuantImpl.apply(tensor, config)


def PPQFloatingQuantFunction(tensor: torch.Tensor, config) -> torch.Tensor:
    """qfunction/floating.py:95-120."""
    if not tensor.is_cuda:
        raise PermissionError('PPQ Floating Quant Function requires tensor device to be cuda, '
                              'CPU floating quantization is not implemented yet.')
    if not _activated(config): return tensor
    if not config.policy.has_property(P.FLOATING):
        raise ValueError('Critical Quantization Error! Unexpected policy detected. '
                         'PPQFloatingQuantFunction except a Floating Quantization Config.')
    if config.policy.has_property(P.DYNAMIC):
        raise ValueError('Unexpected Dynamic Flag in Quantization Policy.')
    if config.policy.has_property(P.PER_CHANNEL):
        return ChannelwiseFloatingQuantImpl.apply(tensor, config.scale, config.offset, config.channel_axis,
                                                  config.exponent_bits, config.mantissa_bits, config.quant_min,
                                                  config.quant_max, config.rounding)
    elif config.policy.has_property(P.PER_TENSOR):
        return TensorwiseFloatingQuantImpl.apply(tensor, _as_1d(config.scale), _as_1d(config.offset),
                                                 config.exponent_bits, config.mantissa_bits, config.quant_min,
                                                 config.quant_max, config.rounding)


def PPQuantFunction(tensor: torch.Tensor, config) -> torch.Tensor:
    """qfunction/__init__.py:10-44."""
    if tensor is None: raise ValueError('Tensor is empty.')
    if config.policy.has_property(P.LINEAR):
        if not config.policy.has_property(P.DYNAMIC):
            return PPQLinearQuantFunction(tensor, config)
        else: return PPQDyamicLinearQuantFunction(tensor, config)
    if config.policy.has_property(P.FLOATING):
        return PPQFloatingQuantFunction(tensor, config)
    raise ValueError('Unexpected Quantization Property Found in PPQuantFunction. '
                     'Do not konw how to quantize your config yet.')


def PPQLinearQuant_toInt(tensor: torch.Tensor, config) -> torch.Tensor:
    """qfunction/linear.py:218-238: quantise only, integer output (int8 / uint8 for 8 bits, int32 above).  The
    reference evaluates clamp(ppq_tensor_round(x / s) + o, qmin, qmax) with torch ops in float32 and casts; here the same
    float32 arithmetic is one HIP kernel (ppqhip_to_int_t / _c), no intermediate tensors."""
    if not config.policy.has_property(P.LINEAR):
        raise ValueError('Critical Quantization Error! Non-linear config detected.')
    if config.num_of_bits == 8:
        if config.policy.has_property(P.SYMMETRICAL): dtype = torch.int8
        elif config.policy.has_property(P.ASYMMETRICAL): dtype = torch.uint8
        else: return None                                   # (the reference falls off its if-chain here too)
    elif config.num_of_bits > 8: dtype = torch.int32
    else: raise Exception('Do not konw how to convert value into int. num of bits is unexpected.')
    if config.policy.has_property(P.PER_CHANNEL): axis = config.channel_axis
    elif config.policy.has_property(P.PER_TENSOR): axis = None
    else: return tensor.type(dtype=dtype)                   # neither granularity: the reference casts the tensor as it is
    if rounding_value(config.rounding) == 5:                # ROUND_TO_NEAR_INT: ppq_tensor_round raises this (utils/round.py:42-43)
        raise NotImplementedError('Torch Tensor can not use this rounding policy(ROUND_TO_NEAR_INT) try ROUND_HALF_EVEN instead.')
    scale, offset = config.scale, config.offset
    if tensor.is_cuda:
        dev = tensor.device
        if scale.device != dev: scale = scale.to(dev)       # the reference's torch expression takes scale / offset from anywhere
        if offset.device != dev: offset = offset.to(dev)
        return CUDA.LinearQuantize_ToInt(tensor, scale, offset, config.quant_min, config.quant_max,
                                         rounding_value(config.rounding), axis, dtype)
    # A host-side tensor (an exporter converting stored weights): the reference evaluates its torch expression wherever the
    # tensor lives.  There is no CPU implementation in this package -- the tensor is staged through the device, converted by
    # the same kernel, and the integers come back to the host; without a device this raises like every other entry point.
    if not torch.cuda.is_available():
        raise RuntimeError('PPQLinearQuant_toInt: the tensor is on the host and no HIP device is available '
                           '(ppq_amd has no CPU path by design)')
    dev = scale.device if scale.is_cuda else torch.device('cuda', torch.cuda.current_device())
    out = CUDA.LinearQuantize_ToInt(tensor.to(dev), scale.to(dev), offset.to(dev), config.quant_min, config.quant_max,
                                    rounding_value(config.rounding), axis, dtype)
    return out.to(tensor.device)


def PPQuantFunction_toInt(tensor: torch.Tensor, config) -> torch.Tensor:
    """qfunction/__init__.py:47-60."""
    if config.policy.has_property(P.LINEAR) and not config.policy.has_property(P.DYNAMIC):
        return PPQLinearQuant_toInt(tensor, config)
    raise ValueError('Unexpected Quantization Property Found in PPQuantFunction_toInt. '
                     'Do not konw how to quantize your config yet.')
