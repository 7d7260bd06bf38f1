"""Quantization data types the hot path is parameterised by -- a host-side mirror of the parts of
``ppq.core`` (reference: ppq/core/quant.py, ppq/core/common.py, ppq/core/config.py) that the
fake-quant functions, observers and the calibration pass read.

These classes are *contracts*, not re-implementations of PPQ's graph machinery: every consumer in
``ppq_amd`` duck-types on the fields below (``policy.has_property``, ``state``, ``scale``,
``offset``, ``quant_min/max``, ``rounding``, ``channel_axis``, ``exponent_bits``, ``mantissa_bits``,
``detail``, ``observer_algorithm``) and compares enum members by ``.value``, so PPQ's own
``TensorQuantizationConfig`` objects work unchanged when this package is dropped into PPQ.
"""
from enum import Enum
from typing import Any

import torch

# ppq/core/common.py:10-34
OBSERVER_MIN_SCALE = 1e-8
OBSERVER_MIN_SCALE_MANUL_OVERRIDE = 'OBSERVER_MIN_SCALE_MANUL_OVERRIDE'
OBSERVER_WARNING = False
OBSERVER_KL_HIST_BINS = 4096
OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE = 'OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE'
OBSERVER_PERCENTILE = 0.9999
OBSERVER_PERCENTILE_MANUL_OVERRIDE = 'OBSERVER_PERCENTILE_MANUL_OVERRIDE'
OBSERVER_MSE_HIST_BINS = 2048
OBSERVER_MSE_COMPUTE_INTERVAL = 8
OBSERVER_FLOATING_MSE_FETCHES = 4096
OBSERVER_ISOTONE_OBSERVER_AXIS = 'OBSERVER_ISOTONE_OBSERVER_AXIS'      # ppq/core/common.py:34


class _Config:
    """ppq/core/config.py:1-21.  The HIP kernels are the only implementation here, hence True."""
    USING_CUDA_KERNEL = True
    NAME = 'ppq_amd (MI355X-native PPQ hot path)'
    VERSION = '0.1.0'
    PPQ_DEBUG = False


PPQ_CONFIG = _Config()


class RoundingPolicy(Enum):
    """ppq/core/quant.py:123-142 (C values: ppq/csrc/cuda/common.cuh:17-24)."""
    ROUND_HALF_EVEN = 0
    ROUND_HALF_UP = 1
    ROUND_HALF_DOWN = 2
    ROUND_HALF_TOWARDS_ZERO = 3
    ROUND_HALF_FAR_FORM_ZERO = 4
    ROUND_TO_NEAR_INT = 5
    ROUND_UP = 6


class QuantizationProperty(Enum):
    """ppq/core/quant.py:145-207."""
    PER_TENSOR = 0x00000001
    PER_CHANNEL = 0x00000002
    LINEAR = 0x00000004
    FLOATING = 0x00000008
    SYMMETRICAL = 0x00000010
    ASYMMETRICAL = 0x00000020
    POWER_OF_2 = 0x00000040
    DYNAMIC = 0x00000080

    def __or__(self, other: int) -> int: return self.value + int(getattr(other, 'value', other))
    def __ror__(self, other: int) -> int: return self.value + int(getattr(other, 'value', other))
    def __add__(self, other: int) -> int: return self.value + int(getattr(other, 'value', other))
    def __radd__(self, other: int) -> int: return self.value + int(getattr(other, 'value', other))


class QuantizationPolicy:
    """ppq/core/quant.py:210-298 (bitmap of QuantizationProperty; validity table :256-287)."""
    def __init__(self, policy: int) -> None:
        P = QuantizationProperty
        p = int(policy)
        granularity = p & (P.PER_TENSOR.value | P.PER_CHANNEL.value)
        kind = p & (P.LINEAR.value | P.FLOATING.value)
        sym = p & (P.SYMMETRICAL.value | P.ASYMMETRICAL.value)
        ok = (granularity in (P.PER_TENSOR.value, P.PER_CHANNEL.value)
              and kind in (P.LINEAR.value, P.FLOATING.value)
              and sym in (P.SYMMETRICAL.value, P.ASYMMETRICAL.value))
        if kind == P.FLOATING.value:   # only SYMMETRICAL | FLOATING | POWER_OF_2 is valid in PPQ
            ok = ok and sym == P.SYMMETRICAL.value and bool(p & P.POWER_OF_2.value) and not (p & P.DYNAMIC.value)
        if not ok:
            raise ValueError('invalid quantization pattern, valid partterns are listed in '
                             'ppq.core.OperationQuantizationPolicy.__check_valid')
        self._policy = p

    def has_property(self, property: QuantizationProperty) -> bool:
        return (self._policy & property.value) != 0

    def to_dict(self) -> dict:
        """quant.py:298-306: property name -> present."""
        return {prop.name: self.has_property(prop) for prop in QuantizationProperty}

    def __eq__(self, o: object) -> bool:
        return isinstance(o, QuantizationPolicy) and self._policy == o._policy

    def __hash__(self) -> int:
        return hash(self._policy)


class QuantizationStates(Enum):
    """ppq/core/quant.py:301-364."""
    INITIAL = 1
    ACTIVATED = 4
    BAKED = 2
    OVERLAPPED = 3
    PASSIVE_INIT = 6
    PASSIVE = 5
    PASSIVE_BAKED = 7
    FP32 = 8
    SOI = -1                  # legacy names the reference keeps (quant.py:352-354); nothing on this path sets them
    DEQUANTIZED = -2
    DEACTIVED = -3

    @ classmethod
    def is_activated(cls, state) -> bool:
        return state_value(state) in (cls.ACTIVATED.value, cls.PASSIVE.value)

    @ classmethod
    def can_export(cls, state) -> bool:
        """quant.py:361-364."""
        return state_value(state) not in (cls.INITIAL.value, cls.PASSIVE_INIT.value, cls.DEQUANTIZED.value, cls.DEACTIVED.value)


class QuantizationVisibility(Enum):
    """ppq/core/quant.py:20-23."""
    FORCE_EXPORT = 1
    EXPORT_WHEN_ACTIVE = 2
    INTERNAL = 3


def state_value(state) -> int:
    return int(getattr(state, 'value', state))


def rounding_value(rounding) -> int:
    return int(getattr(rounding, 'value', rounding))


def is_initial(config) -> bool:
    return state_value(config.state) == QuantizationStates.INITIAL.value


def set_activated(config) -> None:
    """config.state = ACTIVATED, using the enum class the config's current state comes from."""
    cls = type(config.state)
    config.state = cls.ACTIVATED if hasattr(cls, 'ACTIVATED') else QuantizationStates.ACTIVATED


class TensorQuantizationConfig:
    """The fields of ppq.core.TensorQuantizationConfig (ppq/core/quant.py:367-896) that the hot
    path reads or writes, including the dominance union-find (``dominated_by`` / ``master_by``, quant.py:596-713): scale
    and offset of an OVERLAPPED or PASSIVE config resolve through its root, as the reference's properties do."""
    _counter = 0

    def __init__(self, policy: QuantizationPolicy, rounding: RoundingPolicy = RoundingPolicy.ROUND_HALF_EVEN,
                 num_of_bits: int = 8, quant_min: int = -127, quant_max: int = 128, exponent_bits: int = 0,
                 scale: Any = None, offset: Any = None, observer_algorithm: str = None, detail: Any = None,
                 channel_axis: int = None, state: QuantizationStates = QuantizationStates.INITIAL,
                 visibility: 'QuantizationVisibility' = QuantizationVisibility.EXPORT_WHEN_ACTIVE):
        assert 2 <= num_of_bits <= 32, 'Cannot quantize a tensor with less than 2 or more than 32 bits.'
        assert 0 <= exponent_bits <= 8, 'Exponent bits must be in [0, 8].'
        self.policy = policy
        self.exponent_bits = exponent_bits
        self.num_of_bits = num_of_bits
        self._scale = scale
        self._offset = offset
        self.state = state
        self.rounding = rounding
        self.quant_min = quant_min
        self.quant_max = quant_max
        self.channel_axis = channel_axis
        self.observer_algorithm = observer_algorithm
        self.detail = {} if detail is None else detail
        self._dominator = self                               # union-find root pointer (quant.py:596)
        self.visibility = visibility
        TensorQuantizationConfig._counter += 1
        self._hash = TensorQuantizationConfig._counter

    @ property
    def mantissa_bits(self) -> int:
        return self.num_of_bits - self.exponent_bits - 1     # quant.py:794-800

    @ property
    def dominated_by(self):
        """Root of this config's union-find tree (quant.py:647-675), with path compression."""
        if self._dominator is self: return self
        root = self._dominator.dominated_by
        self._dominator = root
        return root

    @ dominated_by.setter
    def dominated_by(self, o) -> None:
        """quant.py:677-691: the trees of self and o are joined under o's root; this config becomes OVERLAPPED.  Refused like
        the reference refuses: a non-config trips its ``assert`` (AssertionError), self-domination and a config whose root is
        this very config (the son would dominate its father) raise ValueError."""
        if not isinstance(o, TensorQuantizationConfig):
            raise AssertionError('Can only set this attribute with another tensor config.')
        if o._hash == self._hash: raise ValueError('Error with TQC.dominated_by = o: o must not equal to TQC its self.')
        root, dominator = self.dominated_by, o.dominated_by
        if dominator is self:
            raise ValueError('Can not Assign Dominator like this, Circular reference was detected. '
                             'Son TQC can not dominate its Father.')
        if root is not dominator:
            root._dominator = dominator
            self._dominator = dominator
            root.state = QuantizationStates.OVERLAPPED
            self.state = QuantizationStates.OVERLAPPED

    @ property
    def master_by(self):
        return self.dominated_by

    @ master_by.setter
    def master_by(self, master) -> None:
        """quant.py:702-713: a PASSIVE config takes scale / offset from its master (Clip bounds, Pad value)."""
        if not isinstance(master, TensorQuantizationConfig):
            raise TypeError('Error with TQC.master_by(o): o must be another Tensor Quantization Config, '
                            f'however {type(master)} was given.')
        if master._hash == self._hash: raise ValueError('Error with TQC.dominated_by = o: o must not equal to TQC its self.')
        self._dominator = master
        self.state = QuantizationStates.PASSIVE if (master.scale is not None and master.offset is not None) \
            else QuantizationStates.PASSIVE_INIT

    @ property
    def scale(self) -> torch.Tensor:
        root = self.dominated_by
        return self._scale if root is self else root.scale     # quant.py:743-749

    @ scale.setter
    def scale(self, value: Any): self._scale = value

    @ property
    def offset(self) -> torch.Tensor:
        root = self.dominated_by
        return self._offset if root is self else root.offset

    @ offset.setter
    def offset(self, value: Any): self._offset = value

    def can_export(self, export_overlapped: bool = False) -> bool:
        """quant.py:601-613: does an exporter write this config?  (``EXPORT_OVERLAPPED_CONFIG`` is False, common.py:115.)"""
        if self.visibility == QuantizationVisibility.INTERNAL: return False
        live = state_value(self.state) in (QuantizationStates.ACTIVATED.value, QuantizationStates.PASSIVE.value,
                                           QuantizationStates.BAKED.value, QuantizationStates.PASSIVE_BAKED.value)
        if export_overlapped and state_value(self.state) == QuantizationStates.OVERLAPPED.value: live = True
        if not (live or self.visibility == QuantizationVisibility.FORCE_EXPORT): return False
        return isinstance(self.scale, torch.Tensor) and isinstance(self.offset, torch.Tensor)

    def is_same_scheme(self, o: object) -> bool:
        """quant.py:634-644: same grid (range, policy, bit widths, channel axis, rounding), whatever the scales are."""
        if not isinstance(o, TensorQuantizationConfig):
            raise TypeError('Can only compare TensorQuantizationConfig object with another TensorQuantizationConfig object.')
        mine = (self.quant_max, self.quant_min, self.policy, self.num_of_bits, self.exponent_bits, self.channel_axis)
        theirs = (o.quant_max, o.quant_min, o.policy, o.num_of_bits, o.exponent_bits, o.channel_axis)
        return mine == theirs and rounding_value(self.rounding) == rounding_value(o.rounding)

    def is_revisable(self) -> bool:
        """quant.py:714-723: its own root, and in a state a pass may still rewrite."""
        S = QuantizationStates
        return self.dominated_by is self and state_value(self.state) in (
            S.ACTIVATED.value, S.FP32.value, S.INITIAL.value, S.PASSIVE.value, S.PASSIVE_INIT.value)

    def copy(self) -> 'TensorQuantizationConfig':
        """quant.py:865-896: a new config (new identity) with the same fields; tensors cloned, ``detail`` copied one level
        deep, an OVERLAPPED copy keeps pointing at the original's dominator."""
        def dup(t): return t.clone() if isinstance(t, torch.Tensor) else t
        twin = TensorQuantizationConfig(policy=self.policy, rounding=self.rounding, num_of_bits=self.num_of_bits,
                                        quant_min=self.quant_min, quant_max=self.quant_max, exponent_bits=self.exponent_bits,
                                        scale=dup(self.scale), offset=dup(self.offset), observer_algorithm=self.observer_algorithm,
                                        detail=dict(self.detail), channel_axis=self.channel_axis, state=self.state,
                                        visibility=self.visibility)
        if state_value(self.state) == QuantizationStates.OVERLAPPED.value: twin._dominator = self._dominator
        return twin

    def __hash__(self) -> int: return self._hash
    def __eq__(self, o: object) -> bool: return isinstance(o, TensorQuantizationConfig) and o._hash == self._hash
    def __str__(self) -> str: return f'ppq_amd TensorQuantizationConfig({self._hash})'


def LinearQuantizationConfig(symmetrical: bool = True, dynamic: bool = False, power_of_2: bool = False,
                             channel_axis: int = None, quant_min: int = -128, quant_max: int = 127,
                             num_of_bits: int = 8, calibration: str = 'minmax',
                             rounding: RoundingPolicy = RoundingPolicy.ROUND_HALF_EVEN) -> TensorQuantizationConfig:
    """ppq.lib.LinearQuantizationConfig, ppq/lib/quant.py:106-134 (same argument order and meaning)."""
    P = QuantizationProperty
    p = P.LINEAR.value
    p += P.SYMMETRICAL.value if symmetrical else P.ASYMMETRICAL.value
    p += P.PER_TENSOR.value if channel_axis is None else P.PER_CHANNEL.value
    if power_of_2: p += P.POWER_OF_2.value
    if dynamic: p += P.DYNAMIC.value
    return TensorQuantizationConfig(policy=QuantizationPolicy(p), rounding=rounding, num_of_bits=num_of_bits,
                                    quant_min=quant_min, quant_max=quant_max, observer_algorithm=calibration,
                                    channel_axis=channel_axis)


def FloatingQuantizationConfig(symmetrical: bool = True, power_of_2: bool = True, channel_axis: int = None,
                               quant_min: float = -448.0, quant_max: float = 448.0, exponent: int = 4,
                               mantissa: int = 3, calibration: str = 'constant',
                               rounding: RoundingPolicy = RoundingPolicy.ROUND_HALF_EVEN) -> TensorQuantizationConfig:
    """ppq.lib.FloatingQuantizationConfig, ppq/lib/quant.py:137-166."""
    P = QuantizationProperty
    p = P.FLOATING.value + (P.SYMMETRICAL.value if symmetrical else P.ASYMMETRICAL.value)
    p += P.PER_CHANNEL.value if channel_axis is not None else P.PER_TENSOR.value
    if power_of_2: p += P.POWER_OF_2.value
    return TensorQuantizationConfig(policy=QuantizationPolicy(p), rounding=rounding,
                                    num_of_bits=exponent + mantissa + 1, exponent_bits=exponent,
                                    quant_min=quant_min, quant_max=quant_max, observer_algorithm=calibration,
                                    channel_axis=channel_axis)
