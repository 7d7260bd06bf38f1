"""Host-side rounding helpers the scale/offset derivations depend on.

Mirror of ppq/utils/round.py (same names, same semantics): ``ppq_numerical_round`` (Decimal
arithmetic on the exact double, :51-95), ``ppq_round_to_power_of_2`` (:115-135) and
``ppq_tensor_round`` (torch ops, :9-49 / :97-113).  Known answers: tests/test_rounding.py of the
reference.
"""
from decimal import ROUND_HALF_DOWN, ROUND_HALF_EVEN, ROUND_HALF_UP, Decimal
from math import ceil, floor, log2
from typing import Union

import torch

from .core import RoundingPolicy, rounding_value

_R = RoundingPolicy


def ppq_numerical_round(value: float, policy: RoundingPolicy = RoundingPolicy.ROUND_HALF_EVEN) -> int:
    assert isinstance(value, float), 'numerical round only takes effect on float number.'
    p = rounding_value(policy)
    if p == _R.ROUND_HALF_EVEN.value:
        return int(Decimal(value).quantize(exp=Decimal(1), rounding=ROUND_HALF_EVEN))
    elif p == _R.ROUND_HALF_UP.value:
        if value > 0: return int(Decimal(value).quantize(exp=Decimal(1), rounding=ROUND_HALF_UP))
        else: return int(Decimal(value).quantize(exp=Decimal(1), rounding=ROUND_HALF_DOWN))
    elif p == _R.ROUND_HALF_DOWN.value:
        if value > 0: return int(Decimal(value).quantize(exp=Decimal(1), rounding=ROUND_HALF_DOWN))
        else: return int(Decimal(value).quantize(exp=Decimal(1), rounding=ROUND_HALF_UP))
    elif p == _R.ROUND_HALF_TOWARDS_ZERO.value:
        return ppq_numerical_round(value, _R.ROUND_HALF_DOWN)
    elif p == _R.ROUND_HALF_FAR_FORM_ZERO.value:
        return ppq_numerical_round(value, _R.ROUND_HALF_UP)
    elif p == _R.ROUND_TO_NEAR_INT.value:
        if value > 0: return floor(value + 0.5)
        else: return ceil(value - 0.5)
    elif p == _R.ROUND_UP.value:
        return ceil(value)
    raise ValueError('Unexpected rounding policy found.')


def ppq_round_to_power_of_2(value: Union[float, int], policy: RoundingPolicy = RoundingPolicy.ROUND_UP) -> float:
    if value == 0: return 0
    sign = 1 if value >= 0 else -1
    assert isinstance(value, (float, int)), 'power-of-2 round only takes effect on float or int.'
    return sign * float(pow(2, ppq_numerical_round(log2(sign * value), policy=policy)))


def ppq_tensor_round(value: torch.Tensor, policy: RoundingPolicy = RoundingPolicy.ROUND_HALF_EVEN) -> torch.Tensor:
    assert isinstance(value, torch.Tensor), 'tensor round only takes effect on torch tensor.'
    p = rounding_value(policy)
    if p == _R.ROUND_HALF_EVEN.value: return value.round()
    elif p == _R.ROUND_UP.value: return value.ceil()
    elif p == _R.ROUND_HALF_TOWARDS_ZERO.value: return torch.sign(value) * torch.ceil(value.abs() - 0.5)
    elif p == _R.ROUND_HALF_FAR_FORM_ZERO.value: return torch.sign(value) * torch.floor(value.abs() + 0.5)
    elif p == _R.ROUND_HALF_DOWN.value: return torch.ceil(value - 0.5)
    elif p == _R.ROUND_HALF_UP.value: return torch.floor(value + 0.5)
    elif p == _R.ROUND_TO_NEAR_INT.value:
        raise NotImplementedError(f'Torch Tensor can not use this rounding policy({policy}) try ROUND_HALF_EVEN instead.')
    raise ValueError('Unexpected rounding policy found.')
