"""Cases + seeded inputs of tests/golden/dynamic.npz (the reference's PPQDyamicLinearQuantFunction, qfunction/linear.py:99-198):
imported by make_golden.py (which runs the REFERENCE on them) and by the tests.  No reference import: travels to the GPU box."""
import torch


def dynamic_cases():
    """(key, shape, channel_axis or None, symmetrical, quant_min, quant_max, power_of_2)."""
    return [
        ('t_asym_u8', (2, 3, 16, 16), None, False, 0, 255, False),
        ('t_sym_s8', (2, 3, 16, 16), None, True, -128, 127, False),
        ('t_sym_s4', (5, 77), None, True, -8, 7, False),
        ('c_sym_s8_a1', (2, 12, 9, 7), 1, True, -128, 127, False),
        ('c_asym_u8_a1', (2, 12, 9, 7), 1, False, 0, 255, False),
        ('c_sym_s8_a0', (16, 8, 3, 3), 0, True, -128, 127, False),
        ('c_asym_u8_last', (4, 10, 24), 2, False, 0, 255, False),
        ('c_sym_s4_a0', (32, 6, 3, 3), 0, True, -8, 7, False),
        ('c_sym_pow2_a1', (3, 8, 11), 1, True, -128, 127, True),
    ]


def dynamic_input(key, shape, axis):
    g = torch.Generator().manual_seed(sum(key.encode()))
    x = torch.randn(shape, generator=g) * 2.5 + 0.3
    if axis is not None:                                   # channels of very different ranges, one all-positive, one constant
        C = shape[axis]
        view = [1] * len(shape); view[axis] = C
        x = x * (torch.rand(C, generator=g) * 4 + 0.05).view(view)
        idx = [slice(None)] * len(shape)
        idx[axis] = 1; x[tuple(idx)] = x[tuple(idx)].abs()
        idx[axis] = 2; x[tuple(idx)] = 0.75
    return x.contiguous()
