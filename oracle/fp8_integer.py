"""A SECOND, independent derivation of the reference's FP8 / (E, M) float simulation -- TEST INFRASTRUCTURE ONLY.

``QuantizeScalarFloating`` (ppq/csrc/cuda/common.cuh:154-226) has no CPU twin and no test in the
reference; ``oracle/ppq_oracle.c`` restates it statement by statement in C (floats, unions, the
rounding-helper trick).  This file derives the SAME function a different way -- from what those
statements mean, on the raw IEEE-754 bit pattern with integer arithmetic only (torch int64 ops, so it runs
on the CPU and, for the exhaustive 2^32 sweep, on the GPU) -- for the ROUND_HALF_EVEN policy every shipped
quantizer uses (FP8Quantizer.py:99,196).  Two derivations that agree on every one of the 2^32 inputs pin
each other; tests/test_oracle_golden.py checks this file against the C restatement, tests/test_gpu_kernels.py
sweeps the HIP kernel against it exhaustively.

Meaning of the reference code, for u = value / scale (bits b, sign s, biased exponent e, mantissa m):

1. saturate (common.cuh:172-185): T = the float with exponent 2^(E-1) and the top M mantissa bits set
   ("theoretical maximum"); hi = min(clip_max, T), lo = max(clip_min, -T); u > hi -> hi, u < lo -> lo
   (both comparisons are false for NaN, which then runs through the bit surgery below like any other pattern).
2. subnormal range (common.cuh:203-208): e - 127 < -(2^(E-1)) + 2  ->  nearbyint(u / q) * q with the quantum
   q = 2^-(2^(E-1) + M - 2): the significand is shifted right and rounded half to even, an exact integer
   operation (the result is converted to int32 first: `_round2int` returns int, common.cuh:88-114).
3. normal range (common.cuh:210-223): keep the top M mantissa bits; the dropped low bits L (D = 23 - M of
   them) round UP only when L > 2^(D-1).  An exact tie rounds DOWN in magnitude, whatever the kept bits are:
   the code rounds the fraction L / 2^D in [0, 1) with nearbyint, and nearbyint(0.5) = 0.  The increment is
   ADDED to the mantissa field, so a carry walks into the exponent (1.1111|1xx -> 2.0).
4. CLIP to [clip_min, clip_max] (common.cuh:225), then dequantise (q - offset) * scale (floating.cu:50-52).
"""
import torch


def _as_bits(x: torch.Tensor) -> torch.Tensor:
    return x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF


def _from_bits(b: torch.Tensor) -> torch.Tensor:
    b = b & 0xFFFFFFFF
    b = torch.where(b >= 0x80000000, b - 0x100000000, b)        # to signed 32-bit range
    return b.to(torch.int32).view(torch.float32)


def theoretical_maximum(exponent: int, mantissa: int) -> float:
    emax = 1 << (exponent - 1)
    bits = ((emax + 127) << 23) + ((~(0x007FFFFF >> mantissa)) & 0x007FFFFF)
    return float(_from_bits(torch.tensor([bits], dtype=torch.int64))[0])


def quantize_unscaled(u: torch.Tensor, exponent: int = 4, mantissa: int = 3, clip_min: float = -448.0,
                      clip_max: float = 448.0) -> torch.Tensor:
    """u (float32, = value / scale) -> the (E, M) float the reference produces, ROUND_HALF_EVEN."""
    assert u.dtype == torch.float32
    E, M = int(exponent), int(mantissa)
    D = 23 - M
    T = theoretical_maximum(E, M)
    hi, lo = min(float(clip_max), T), max(float(clip_min), -T)
    b = _as_bits(u)
    sign = b & 0x80000000
    e = (b >> 23) & 0xFF
    m = b & 0x007FFFFF
    # ---- normal range: integer rounding of the mantissa field, carry into the exponent by addition
    low = m & ((1 << D) - 1)
    up = (low > (1 << (D - 1))).to(torch.int64)                      # strictly more than half: ties go DOWN
    kept = ((m >> D) + up) << D
    normal = _from_bits(sign + kept + (e << 23))
    normal = torch.where(normal > clip_max, torch.full_like(normal, clip_max), normal)      # CLIP: NaN passes through
    normal = torch.where(normal < clip_min, torch.full_like(normal, clip_min), normal)
    # ---- subnormal range: round the significand to a multiple of q = 2^-K, half to even, as an integer
    K = (1 << (E - 1)) + M - 2
    sig = torch.where(e > 0, m | 0x00800000, m)                      # |u| = sig * 2^(max(e,1) - 150)
    shift = 150 - torch.clamp(e, min=1) - K                          # |u| / q = sig >> shift   (shift >= 1 here)
    shift = torch.clamp(shift, min=1, max=62)
    whole = sig >> shift
    rem = sig & ((torch.ones_like(sig) << shift) - 1)
    half = torch.ones_like(sig) << (shift - 1)
    r = whole + ((rem > half) | ((rem == half) & ((whole & 1) == 1))).to(torch.int64)
    sub = r.to(torch.float32) * (2.0 ** -K)                          # exact: r < 2^(M+2)
    sub = torch.where((sign != 0) & (r != 0), -sub, sub)             # the int32 round trip loses the sign of zero
    out = torch.where((e - 127) < (-(1 << (E - 1)) + 2), sub, normal)
    # ---- saturation first (the reference returns before any bit surgery)
    out = torch.where(u > hi, torch.full_like(out, hi), out)
    out = torch.where(u < lo, torch.full_like(out, lo), out)
    return out


def fq_float_t(x: torch.Tensor, scale: float, offset: float = 0.0, exponent: int = 4, mantissa: int = 3,
               clip_min: float = -448.0, clip_max: float = 448.0) -> torch.Tensor:
    """floating.cu:36-55 for one scale: dequant(quant(x / s)).  The division is float32 IEEE; with a
    power-of-two scale it is exact in every implementation."""
    s = torch.tensor(scale, dtype=torch.float32, device=x.device)
    o = torch.tensor(offset, dtype=torch.float32, device=x.device)
    q = quantize_unscaled(x / s, exponent, mantissa, clip_min, clip_max)
    return (q - o) * s
