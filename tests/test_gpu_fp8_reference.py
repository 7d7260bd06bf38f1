"""GPU: the HIP kernels against outputs PRODUCED BY THE REFERENCE'S OWN SOURCE.

``ppq/csrc/cuda/common.cuh`` -- ``QuantizeScalarFloating``, ``_round2int``, ``QuantizeScalar``, ``DequantizeScalar`` -- is
header-only device code; ``oracle/Makefile`` compiles it as host C++ where it lies (``oracle/_ref/libref_common.so``).
``tests/golden/fp8_ref.npz`` holds what it returned on the structured sweep (committed; generator
``tests/golden/make_golden.py::gen_fp8_ref``), and the ``.so`` itself travels to the GPU box for the larger sweeps.
Everything here is bit-exact; NaN payloads aside (a NaN must stay a NaN)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_common as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def CUDA():
    from ppq_amd import CUDA as C
    return C


def _same(a: np.ndarray, b: np.ndarray):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(DEV)


def test_fp8_kernel_equals_reference_goldens(CUDA, golden_dir):
    """FloatingQuantize_T on the 26 golden cases: E4M3 / E5M2 x {4 scales RNE, the 7 other rounding modes, a raw float
    offset, a clip wider than the format's theoretical maximum}."""
    z = np.load(os.path.join(golden_dir, 'fp8_ref.npz'))
    n = 0
    for key, fmt, scale, offset, clip, rounding in R.fp8_ref_cases():
        E, M, c = R.FORMATS[fmt]
        if clip is not None: c = clip
        x = R.sweep_bits(M, n_random=8192, seed=0)
        got = CUDA.FloatingQuantize_T(_dev(x), _dev([scale]), _dev([offset]), E, M, -c, c, rounding).cpu().numpy()
        want = z[key].view(np.float32)
        same = _same(got, want)
        assert same.all(), (key, [(hex(int(x.view(np.uint32)[i])), got[i], want[i]) for i in np.nonzero(~same)[0][:5]])
        n += x.size
    assert n > 600_000


def test_device_rounding_modes_equal_reference_goldens(CUDA, golden_dir):
    """_round2int on the DEVICE, all 8 modes (row a1): LinearQuantize_T with scale 1, offset 0 and a clip range wider than
    the values is round2int itself; compared with what the reference's common.cuh returned (goldens)."""
    z = np.load(os.path.join(golden_dir, 'fp8_ref.npz'))
    v, want = z['round_values'], z['round_results']
    keep = np.abs(v) < 2.0 ** 23                                   # results exactly representable as float32 on the way back
    one, zero = _dev([1.0]), _dev([0.0])
    for rounding in range(8):
        got = CUDA.LinearQuantize_T(_dev(v[keep]), one, zero, -2 ** 30, 2 ** 30, rounding).cpu().numpy()
        assert np.array_equal(got.astype(np.int64), want[rounding][keep].astype(np.int64)), \
            (rounding, v[keep][got != want[rounding][keep]][:5])
    assert keep.sum() > 4000


needs_ref = pytest.mark.skipif(not R.available(), reason='oracle/_ref/libref_common.so not built (needs /root/reference at build time)')


@needs_ref
@pytest.mark.parametrize('fmt', ['e4m3', 'e5m2', 'e3m4', 'e5m10'])
def test_fp8_kernel_equals_reference_source_large_sweep(CUDA, fmt):
    """The same comparison against the compiled reference source directly: structured sweep + 4 M random bit patterns,
    power-of-two scales (the division-free fast path) and generic ones, all 8 rounding modes, per-tensor and per-channel."""
    formats = dict(R.FORMATS); formats.update({'e3m4': (3, 4, 30.0), 'e5m10': (5, 10, 65504.0)})
    E, M, c = formats[fmt]
    x = R.sweep_bits(M, n_random=4_000_000, seed=11)
    xd = _dev(x)
    for scale in (1.0, 2.0 ** -5, 16.0, 0.3, 7.3e-3):
        for rounding in range(8):
            if fmt not in R.FORMATS and rounding not in (0, 4): continue
            got = CUDA.FloatingQuantize_T(xd, _dev([scale]), _dev([0.0]), E, M, -c, c, rounding).cpu().numpy()
            want = R.fq_float_t(x, [scale], [0.0], E, M, -c, c, rounding)
            same = _same(got, want)
            assert same.all(), (fmt, scale, rounding, [(hex(int(x.view(np.uint32)[i])), got[i], want[i]) for i in np.nonzero(~same)[0][:5]])
    rng = np.random.default_rng(2)
    xc = (rng.standard_normal((6, 24, 9, 5)) * rng.choice([1e-3, 0.1, 3, 90, 4000], (6, 24, 9, 5))).astype(np.float32)
    for axis in (0, 1, 3):
        C = xc.shape[axis]
        s = (2.0 ** rng.integers(-6, 4, C)).astype(np.float32)
        if axis == 1: s = (rng.random(C) * 0.5 + 0.01).astype(np.float32)         # generic scales on one axis
        o = rng.integers(-2, 3, C).astype(np.float32)
        got = CUDA.FloatingQuantize_C(_dev(xc), _dev(s), _dev(o), axis, E, M, -c, c, 0).cpu().numpy()
        assert _same(got, R.fq_float_c(xc, s, o, axis, E, M, -c, c, 0)).all(), (fmt, axis)


@needs_ref
def test_linear_kernel_every_rounding_mode_equals_reference_source(CUDA):
    """LinearQuantize_T vs the reference's QuantizeScalar / DequantizeScalar (common.cuh:116-147) in the kernel body of
    linear.cu:49-57, all 8 rounding modes, fractional offsets (rounded half away there), int8 / uint8 / int4."""
    rng = np.random.default_rng(4)
    x = np.concatenate([(rng.standard_normal(1_000_000) * 40), np.arange(-2048, 2048) / 16.0]).astype(np.float32)
    for rounding in range(8):
        for scale, offset, qmin, qmax in ((0.31, 0.0, -128, 127), (0.05, 127.5, 0, 255), (1.7, -2.5, -8, 7), (0.0625, 3.0, -128, 127)):
            got = CUDA.LinearQuantize_T(_dev(x), _dev([scale]), _dev([offset]), qmin, qmax, rounding).cpu().numpy()
            want = R.fq_linear_t(x, [scale], [offset], qmin, qmax, rounding)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rounding, scale, offset)
