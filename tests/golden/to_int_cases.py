"""Cases + seeded inputs of tests/golden/to_int.npz: imported by make_golden.py (which runs the REFERENCE on them) and by the
tests (which run the oracle and the HIP kernels on them).  No reference import here: this file travels to the GPU box."""
import torch


def to_int_cases():
    """(key, shape, per_channel axis or None, symmetrical, bits, qmin, qmax, rounding value) -- shared with the tests."""
    cases = []
    for r in (0, 1, 2, 3, 4, 6):                 # every policy ppq_tensor_round implements (5 = ROUND_TO_NEAR_INT raises)
        cases.append((f't_s8_r{r}', (3, 5, 7, 9), None, True, 8, -128, 127, r))
        cases.append((f'c_u8_r{r}', (4, 6, 5, 8), 1, False, 8, 0, 255, r))
    cases.append(('c_s8_axis0', (16, 3, 3, 3), 0, True, 8, -128, 127, 0))
    cases.append(('t_i32', (2, 1000), None, True, 32, -2 ** 31, 2 ** 31 - 1, 0))
    cases.append(('c_i16_last', (5, 12), 1, True, 16, -32768, 32767, 0))
    cases.append(('t_s4', (1, 3, 224, 224), None, True, 8, -8, 7, 0))          # 4-bit range in an 8-bit container
    return cases


def to_int_inputs(key, shape, axis):
    """Seeded inputs: values on and around the rounding ties, fractional (LSQ-trained) offsets."""
    g = torch.Generator().manual_seed(sum(key.encode()))
    n = 1
    for d in shape: n *= d
    x = torch.randn(n, generator=g) * 40
    x[::7] = torch.round(x[::7]) + 0.5                 # exact ties (scale 1 channels below)
    x[::11] = torch.round(x[::11]) - 0.5
    x = x.reshape(shape)
    C = 1 if axis is None else shape[axis]
    scale = torch.rand(C, generator=g) * 0.9 + 0.1
    scale[0] = 1.0
    offset = torch.randint(-20, 140, [C], generator=g).float()
    if C > 1: offset[1] += 0.37                        # the raw offset is used, not the rounded one
    else: offset += 0.37 if key.endswith('r0') else 0.0
    return x, scale, offset
