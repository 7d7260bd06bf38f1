"""The reference's OWN scalar device functions, compiled as host C++ -- TEST INFRASTRUCTURE ONLY.

``oracle/_ref/libref_common.so`` is built by ``make -C oracle ref`` from
``/root/reference/ppq/csrc/cuda/common.cuh`` where it lies (``oracle/ref_common_shim.cc`` + the stand-in
headers of ``oracle/ref_host_stubs/``; nothing of the reference is copied into this repository).  It is what
pins ``QuantizeScalarFloating`` (common.cuh:154-226: FP8 / any (E, M) float simulation, CUDA-only in the
reference, no test there) and the 8 modes of ``_round2int`` (common.cuh:88-114) to reference-PRODUCED outputs:

* in the build container ``tests/test_oracle_golden.py`` checks ``oracle/ppq_oracle.c`` against it, and
  ``tests/golden/make_golden.py`` writes ``tests/golden/fp8_ref.npz`` from it;
* on the GPU box the prebuilt ``.so`` travels with the snapshot (like ``libref_hist_mse.so``), and the
  committed golden file is there either way.

``sweep_bits`` is the input set both use.  Host-vs-device caveat (shim header): float -> int32 conversion out of
range is undefined on the host, saturating on the device -- ``sweep_bits`` therefore keeps finite inputs with
|x| < 2^24 for the generic sweep (the saturation branches of the FP8 function return before any conversion, so
large and infinite inputs ARE swept with the clip bounds of the real formats).
"""
import ctypes
import os
from typing import Optional

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libref_common.so')
_lib = None

FORMATS = {'e4m3': (4, 3, 448.0), 'e5m2': (5, 2, 57344.0)}      # FP8Quantizer.py:99,196 (exponent, mantissa, clip)


def available() -> bool:
    return os.path.exists(_PATH)


def lib() -> Optional[ctypes.CDLL]:
    global _lib
    if _lib is None:
        if not available(): return None
        r = ctypes.CDLL(_PATH)
        r.ref_round2int.restype = ctypes.c_int
        r.ref_round2int.argtypes = [ctypes.c_float, ctypes.c_int]
        r.ref_quantize_scalar.restype = ctypes.c_int
        r.ref_quantize_scalar.argtypes = [ctypes.c_float, ctypes.c_float] + [ctypes.c_int] * 4
        r.ref_dequantize_scalar.restype = ctypes.c_float
        r.ref_dequantize_scalar.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int]
        r.ref_quantize_scalar_floating.restype = ctypes.c_float
        r.ref_quantize_scalar_floating.argtypes = [ctypes.c_float] * 3 + [ctypes.c_int] * 2 + [ctypes.c_float] * 2 + [ctypes.c_int]
        _lib = r
    return _lib


def _need():
    r = lib()
    if r is None: raise FileNotFoundError('oracle/_ref/libref_common.so missing (built only where /root/reference exists)')
    return r


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def round2int(value: float, rounding: int) -> int:
    return int(_need().ref_round2int(value, rounding))


def quantize_scalar_floating(value, scale=1.0, offset=0.0, exponent=4, mantissa=3, clip_min=-448.0, clip_max=448.0, rounding=0) -> float:
    return float(_need().ref_quantize_scalar_floating(value, scale, offset, exponent, mantissa, clip_min, clip_max, rounding))


def fq_float_t(x, scale, offset, exponent=4, mantissa=3, clip_min=-448.0, clip_max=448.0, rounding=0) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32); out = np.empty_like(x)
    s = np.ascontiguousarray(scale, np.float32).reshape(-1); o = np.ascontiguousarray(offset, np.float32).reshape(-1)
    _need().ref_fq_float_t(_p(x), ctypes.c_int64(x.size), _p(s), _p(o), ctypes.c_int(exponent), ctypes.c_int(mantissa),
                           ctypes.c_float(clip_min), ctypes.c_float(clip_max), ctypes.c_int(rounding), _p(out))
    return out


def fq_float_c(x, scale, offset, channel_axis, exponent=4, mantissa=3, clip_min=-448.0, clip_max=448.0, rounding=0) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32); out = np.empty_like(x)
    s = np.ascontiguousarray(scale, np.float32).reshape(-1); o = np.ascontiguousarray(offset, np.float32).reshape(-1)
    c = int(x.shape[channel_axis]); epc = int(np.prod(x.shape[channel_axis + 1:], dtype=np.int64))
    _need().ref_fq_float_c(_p(x), ctypes.c_int64(x.size), ctypes.c_int64(epc), ctypes.c_int(c), _p(s), _p(o), ctypes.c_int(exponent),
                           ctypes.c_int(mantissa), ctypes.c_float(clip_min), ctypes.c_float(clip_max), ctypes.c_int(rounding), _p(out))
    return out


def fq_linear_t(x, scale, offset, qmin: int, qmax: int, rounding: int = 0) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32); out = np.empty_like(x)
    s = np.ascontiguousarray(scale, np.float32).reshape(-1); o = np.ascontiguousarray(offset, np.float32).reshape(-1)
    _need().ref_fq_linear_t(_p(x), ctypes.c_int64(x.size), _p(s), _p(o), ctypes.c_int(qmin), ctypes.c_int(qmax),
                            ctypes.c_int(rounding), _p(out))
    return out


def fp8_ref_cases():
    """(key, format, scale, offset, clip, rounding) of tests/golden/fp8_ref.npz -- written by tests/golden/make_golden.py::gen_fp8_ref, read by the tests."""
    cases = []
    for fmt in ('e4m3', 'e5m2'):
        for scale in (1.0, 0.125, 4.0, 0.3):
            cases.append((f'{fmt}_s{scale}_r0', fmt, scale, 0.0, None, 0))
        for rounding in range(1, 8):
            cases.append((f'{fmt}_s1.0_r{rounding}', fmt, 1.0, 0.0, None, rounding))
        cases.append((f'{fmt}_s0.5_o3_r0', fmt, 0.5, 3.0, None, 0))                 # the raw float offset of floating.cu:50-52
        cases.append((f'{fmt}_wideclip_r0', fmt, 1.0, 0.0, 1e30, 0))                # theoretical maximum (common.cuh:169-185)
    return cases


def sweep_bits(mantissa: int, n_random: int = 0, seed: int = 0, limit_exp: Optional[int] = None) -> np.ndarray:
    """float32 inputs (as a float32 array) that walk every branch of QuantizeScalarFloating for a format with
    `mantissa` kept bits: every sign / biased exponent (incl. zero / denormals, inf / NaN) x every kept-mantissa
    pattern x dropped bits at {0, 1, half-1, half (the tie), half+1, all ones}; plus `n_random` random bit
    patterns.  `limit_exp`: drop finite values with |x| >= 2^limit_exp (host int conversion, see the module docstring)."""
    D = 23 - mantissa
    lows = np.array([0, 1, (1 << (D - 1)) - 1, 1 << (D - 1), (1 << (D - 1)) + 1, (1 << D) - 1], dtype=np.uint64)
    hi9 = np.arange(512, dtype=np.uint64)[:, None, None] << np.uint64(23)
    kept = np.arange(1 << mantissa, dtype=np.uint64)[None, :, None] << np.uint64(D)
    bits = (hi9 | kept | lows[None, None, :]).reshape(-1)
    if n_random:
        bits = np.concatenate([bits, np.random.default_rng(seed).integers(0, 2 ** 32, size=n_random, dtype=np.uint64)])
    bits = bits.astype(np.uint32)
    if limit_exp is not None:
        e = (bits >> np.uint32(23)) & np.uint32(0xFF)
        bits = bits[(e < 127 + limit_exp) | (e == 255)]
    return bits.view(np.float32).copy()
