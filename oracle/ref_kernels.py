"""The reference's OWN ``__global__`` kernel bodies, run on the host -- TEST INFRASTRUCTURE ONLY.

``oracle/_ref/libref_kernels.so`` is built by ``make -C oracle ref`` (only where ``/root/reference`` exists):
``extract_kernels.awk`` pulls the text of every ``__global__`` function out of ``ppq/csrc/cuda/{sort,linear,floating}.cu``
where they lie into git-ignored ``oracle/_ref/*_kernels.inc``; ``ref_{sort,linear,floating}_shim.cc`` include that text
and ``ref_kernel_host.h`` executes it thread by thread (blockIdx / threadIdx as plain globals, ``atomicAdd`` as ``+=``,
``BlockReduceSum`` as a per-block running sum).  Nothing of the reference is stored in this repository.

What this pins to reference-PRODUCED outputs (VERDICT r3 "Missing 2": rows a9, a12, a16 rested on restatements):

* ``_Histogram_T`` / ``_Histogram_Asymmetric_T`` / ``_Histogram_C`` (sort.cu:75-89, 113-139, 167-185): the bin rule
  ``floor(|x| / hist_scale)`` resp. ``floor((x - min) / ((max - min) / bins))``, the outlier rule, the channel index;
* ``_Quantile_T`` (sort.cu:6-20): the index rule ``__float2int_rn(n * q)`` incl. the int64 -> float conversion of n
  (``quantile_positions`` observes the two reads for any n < 2^31 without allocating the tensor);
* ``_QuantizeTensor_LT_B`` / ``_LC_B`` (linear.cu:235-282, 326-380): clip masks -> grad_x, the per-element scale
  gradient term, and (to float-summation tolerance) grad_s; plus the four forward kernels incl. the vectorised ones;
* ``_QuantizeTensor_FT_B`` / ``_FC_B`` (floating.cu:133-182, 223-283) and the FP8 forward kernels.

The functions mirror ``oracle/ppq_oracle.py``'s signatures so a test can run both on the same arguments.
Host-vs-device caveat: ``int b = floor(...)`` of a value outside int32 (or inf / NaN) is undefined in host C++ and
saturating on the device: keep ``|x| / hist_scale`` below 2^31 in comparisons against this library.
"""
import ctypes
import os
from typing import Optional

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libref_kernels.so')
_lib = None
_c = ctypes


def available() -> bool:
    return os.path.exists(_PATH)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not available():
            raise FileNotFoundError('oracle/_ref/libref_kernels.so missing (built only where /root/reference exists)')
        _lib = ctypes.CDLL(_PATH)
        _lib.ref_quantile_positions.restype = _c.c_int
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_c.c_void_p)


def _geometry(shape, channel_axis):
    """(num_of_channel, element_per_channel) as the reference's host functions derive them (sort.cu:204-208,
    linear.cu:213-214: the stride of the channel axis of the contiguous tensor)."""
    c = int(shape[channel_axis]); epc = 1
    for d in shape[channel_axis + 1:]: epc *= int(d)
    return c, epc


# ---- sort.cu -------------------------------------------------------------------------------------------------
def hist_sym_t(x, hist_scale: float, hist: np.ndarray, clip_outliers: bool = True) -> np.ndarray:
    x = _f32(x); assert hist.dtype == np.int32 and hist.flags.c_contiguous
    lib().ref_hist_sym_t(_p(x), _c.c_int64(x.size), _c.c_int64(hist.size), _c.c_float(hist_scale),
                         _c.c_int(bool(clip_outliers)), _p(hist))
    return hist


def hist_asym_t(x, vmin: float, vmax: float, hist: np.ndarray, clip_outliers: bool = True) -> np.ndarray:
    x = _f32(x); assert hist.dtype == np.int32 and hist.flags.c_contiguous
    lib().ref_hist_asym_t(_c.c_float(vmin), _c.c_float(vmax), _p(x), _c.c_int64(x.size), _c.c_int64(hist.size),
                          _c.c_int(bool(clip_outliers)), _p(hist))
    return hist


def hist_sym_c(x, channel_axis: int, hist_scale: float, hist: np.ndarray, clip_outliers: bool = True) -> np.ndarray:
    x = _f32(x); assert hist.dtype == np.int32 and hist.flags.c_contiguous
    c, epc = _geometry(x.shape, channel_axis)
    assert hist.size % c == 0
    lib().ref_hist_sym_c(_p(x), _c.c_int64(x.size), _c.c_int64(epc), _c.c_int(c), _c.c_int64(hist.size // c),
                         _c.c_float(hist_scale), _c.c_int(bool(clip_outliers)), _p(hist))
    return hist


def quantile_t(x, q: float) -> np.ndarray:
    x = _f32(x).reshape(-1); out = np.zeros(2, np.float32)
    lib().ref_quantile_t(_p(x), _c.c_int64(x.size), _c.c_float(q), _p(out))
    return out


def quantile_positions(n: int, q: float):
    """(max_pos, min_pos) that ``_Quantile_T`` reads for a tensor of n elements."""
    pos = np.zeros(2, np.int64)
    rc = lib().ref_quantile_positions(_c.c_int64(n), _c.c_float(q), _p(pos))
    if rc != 0: raise RuntimeError(f'ref_quantile_positions failed ({rc})')
    return int(pos[0]), int(pos[1])


def isotone_t(x) -> np.ndarray:
    x = _f32(x).reshape(-1); out = np.zeros(4, np.float32)
    lib().ref_isotone_t(_p(x), _c.c_int64(x.size), _p(out))
    return out


# ---- linear.cu -----------------------------------------------------------------------------------------------
def fq_linear_t(x, scale, offset, qmin: int, qmax: int, rounding: int = 0) -> np.ndarray:
    x = _f32(x); out = np.empty_like(x)
    lib().ref_kernel_fq_linear_t(_p(x), _c.c_int64(x.size), _p(_f32(scale).reshape(-1)), _p(_f32(offset).reshape(-1)),
                                 _c.c_int(qmin), _c.c_int(qmax), _c.c_int(rounding), _p(out))
    return out


def fq_linear_c(x, scale, offset, channel_axis: int, qmin: int, qmax: int, rounding: int = 0) -> np.ndarray:
    x = _f32(x); out = np.empty_like(x)
    c, epc = _geometry(x.shape, channel_axis)
    lib().ref_kernel_fq_linear_c(_p(x), _c.c_int64(x.size), _c.c_int32(epc), _c.c_int32(c), _p(_f32(scale).reshape(-1)),
                                 _p(_f32(offset).reshape(-1)), _c.c_int(qmin), _c.c_int(qmax), _c.c_int(rounding), _p(out))
    return out


def fq_linear_t_bwd(x, scale, offset, dy, qmin, qmax, rounding=0, with_partials=False):
    x = _f32(x); dy = _f32(dy); gx = np.empty_like(x); gs = np.zeros(1, np.float32)
    part = np.zeros(x.shape, np.float32) if with_partials else None
    lib().ref_kernel_fq_linear_t_bwd(_p(x), _p(dy), _c.c_int64(x.size), _p(_f32(scale).reshape(-1)),
                                     _p(_f32(offset).reshape(-1)), _c.c_int(qmin), _c.c_int(qmax), _c.c_int(rounding),
                                     _p(gx), _p(gs), _p(part))
    return (gx, gs, part) if with_partials else (gx, gs)


def fq_linear_c_bwd(x, scale, offset, dy, channel_axis, qmin, qmax, rounding=0, with_partials=False):
    x = _f32(x); dy = _f32(dy); gx = np.empty_like(x)
    c, epc = _geometry(x.shape, channel_axis)
    gs = np.zeros(c, np.float32)
    part = np.zeros((c, epc), np.float32) if with_partials else None
    lib().ref_kernel_fq_linear_c_bwd(_p(x), _p(dy), _c.c_int64(x.size), _c.c_int32(epc), _c.c_int32(c),
                                     _p(_f32(scale).reshape(-1)), _p(_f32(offset).reshape(-1)), _c.c_int(qmin),
                                     _c.c_int(qmax), _c.c_int(rounding), _p(gx), _p(gs), _p(part))
    return (gx, gs, part) if with_partials else (gx, gs)


# ---- floating.cu ---------------------------------------------------------------------------------------------
def fq_float_t(x, scale, offset, exponent=4, mantissa=3, clip_min=-448.0, clip_max=448.0, rounding=0):
    x = _f32(x); out = np.empty_like(x)
    lib().ref_kernel_fq_float_t(_p(x), _c.c_int64(x.size), _p(_f32(scale).reshape(-1)), _p(_f32(offset).reshape(-1)),
                                _c.c_int(exponent), _c.c_int(mantissa), _c.c_float(clip_min), _c.c_float(clip_max),
                                _c.c_int(rounding), _p(out))
    return out


def fq_float_c(x, scale, offset, channel_axis, exponent=4, mantissa=3, clip_min=-448.0, clip_max=448.0, rounding=0):
    x = _f32(x); out = np.empty_like(x)
    c, epc = _geometry(x.shape, channel_axis)
    lib().ref_kernel_fq_float_c(_p(x), _c.c_int64(x.size), _c.c_int64(epc), _c.c_int(c), _p(_f32(scale).reshape(-1)),
                                _p(_f32(offset).reshape(-1)), _c.c_int(exponent), _c.c_int(mantissa),
                                _c.c_float(clip_min), _c.c_float(clip_max), _c.c_int(rounding), _p(out))
    return out


def fq_float_c_bwd(x, scale, offset, dy, channel_axis, exponent, mantissa, clip_min, clip_max, rounding=0):
    """channel_axis=None -> the per-tensor kernel (_QuantizeTensor_FT_B)."""
    x = _f32(x); dy = _f32(dy); gx = np.empty_like(x)
    s = _f32(scale).reshape(-1); o = _f32(offset).reshape(-1)
    if channel_axis is None:
        gs = np.zeros(1, np.float32)
        lib().ref_kernel_fq_float_t_bwd(_p(x), _p(dy), _c.c_int64(x.size), _p(s), _p(o), _c.c_int(exponent),
                                        _c.c_int(mantissa), _c.c_float(clip_min), _c.c_float(clip_max),
                                        _c.c_int(rounding), _p(gx), _p(gs))
        return gx, gs
    c, epc = _geometry(x.shape, channel_axis)
    gs = np.zeros(c, np.float32)
    lib().ref_kernel_fq_float_c_bwd(_p(x), _p(dy), _c.c_int64(x.size), _c.c_int64(epc), _c.c_int(c), _p(s), _p(o),
                                    _c.c_int(exponent), _c.c_int(mantissa), _c.c_float(clip_min), _c.c_float(clip_max),
                                    _c.c_int(rounding), _p(gx), _p(gs))
    return gx, gs
