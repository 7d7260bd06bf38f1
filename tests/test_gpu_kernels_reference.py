"""GPU: the HIP histogram / quantile / LSQ-backward / FP8-backward kernels against outputs PRODUCED BY THE REFERENCE'S OWN KERNEL
BODIES -- the counterpart of tests/test_gpu_fp8_reference.py for sort.cu / linear.cu / floating.cu.

``tests/golden/kernels_ref.npz`` holds what ``_Histogram_T``, ``_Histogram_Asymmetric_T``, ``_Histogram_C``, ``_Quantile_T``,
``_QuantizeTensor_LT_B / _LC_B`` and ``_QuantizeTensor_FT_B / _FC_B`` returned when their text, extracted from the reference at
build time, ran on the host (oracle/ref_kernels.py); the ``.so`` itself travels to the GPU box for the larger sweeps.
Counts, picks and clip masks are compared bit for bit; scale gradients (float sums) within the stated tolerance."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ref_kernels as K

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import kernel_ref_cases as cases  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda'
needs_ref = pytest.mark.skipif(not K.available(), reason='oracle/_ref/libref_kernels.so not built (needs /root/reference at build time)')


@pytest.fixture(scope='module')
def CUDA():
    from ppq_amd import CUDA as C
    return C


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'kernels_ref.npz'))


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(DEV)


def _hip_hist(CUDA, kind, v, prm):
    if kind == 'sym':
        h = torch.zeros(cases.BINS, dtype=torch.int32, device=DEV)
        CUDA.Histogram_T(_dev(v), h, prm[0], prm[1])
    elif kind == 'asym':
        h = torch.zeros(cases.BINS, dtype=torch.int32, device=DEV)
        CUDA.Histogram_Asymmetric_T(prm[0], prm[1], _dev(v), h, prm[2])
    else:
        shape, axis, hs, clip = prm
        h = torch.zeros(shape[axis], 128, dtype=torch.int32, device=DEV)
        CUDA.Histogram_C(_dev(v), axis, h, hs, clip)
    return h.cpu().numpy().reshape(-1)


def test_histogram_kernels_equal_reference_kernel_goldens(CUDA, gold):
    """Histogram_T / Histogram_Asymmetric_T / Histogram_C on the boundary vectors: every count equals what the reference's
    kernel text produced (the reciprocal-multiply + exactness-test binning of hist.hip must land every value in the bin of
    floor(IEEE quotient))."""
    n = 0
    for key, kind, v, prm in cases.hist_cases():
        got, want = _hip_hist(CUDA, kind, v, prm), gold['hist_' + key]
        assert np.array_equal(got, want), (key, np.nonzero(got != want)[0][:8])
        n += 1
    assert n >= 40


def test_histogram_rows_path_equals_reference_kernel_goldens(CUDA, gold):
    """The persistent-rows path the observers use (one multi-job launch, folded at render) on the same vectors."""
    sym = [(key, v, prm) for key, kind, v, prm in cases.hist_cases() if kind == 'sym' and prm[1]]
    rows = [torch.zeros(CUDA.hist_rows(), cases.BINS, dtype=torch.int32, device=DEV) for _ in sym]
    CUDA.Histogram_T_Rows_Multi([_dev(v) for _, v, _ in sym], rows, [p[0] for _, _, p in sym], True)
    for (key, _, _), r in zip(sym, rows):
        h = torch.zeros(cases.BINS, dtype=torch.int32, device=DEV)
        CUDA.Histogram_Rows_Finish(r, h)
        assert np.array_equal(h.cpu().numpy(), gold['hist_' + key]), key
    asym = [(key, v, prm) for key, kind, v, prm in cases.hist_cases() if kind == 'asym' and prm[2]]
    rows = [torch.zeros(CUDA.hist_rows(), cases.BINS, dtype=torch.int32, device=DEV) for _ in asym]
    CUDA.Histogram_Asymmetric_T_Rows_Multi([p[0] for _, _, p in asym], [p[1] for _, _, p in asym], [_dev(v) for _, v, _ in asym], rows, True)
    for (key, _, _), r in zip(asym, rows):
        h = torch.zeros(cases.BINS, dtype=torch.int32, device=DEV)
        CUDA.Histogram_Rows_Finish(r, h)
        assert np.array_equal(h.cpu().numpy(), gold['hist_' + key]), key


def test_quantile_picks_equal_reference_kernel_goldens(CUDA, gold):
    """Quantile_T == what `_Quantile_T` picked from the sorted copy (small tensors, ties, q = 0 / 1 / .5)."""
    for key, v, q in cases.quantile_arrays():
        got = CUDA.Quantile(_dev(v), q).cpu().numpy()
        assert np.array_equal(got, gold[key]), (key, got, gold[key])


@pytest.mark.parametrize('n', [16777217, 16777219, 33554433, 51380224, 205520896])
def test_quantile_index_rule_at_size_equals_reference_kernel_positions(CUDA, gold, n):
    """n > 2^24, where `num_of_elements * q` rounds n to float first: the HIP result must be the element of a device sort at
    the positions the reference kernel itself read (recorded in the goldens for exactly this n)."""
    i = cases.QUANTILE_NS.index(n)
    g = torch.Generator(device=DEV).manual_seed(n % 1000)
    x = torch.randn(n, device=DEV, generator=g)
    srt = torch.sort(x).values
    for j, q in enumerate(cases.QUANTILE_QS):
        if q not in (0.9999, 0.99999, 0.999, 0.5): continue
        mx, mn = (int(a) for a in gold['qpos'][i, j])
        got = CUDA.Quantile(x, q).cpu().tolist()
        assert got == [srt[mx].item(), srt[mn].item()], (n, q, mx, mn)
    del srt, x


def _factor(n, qmin, qmax, per_channel):
    return float(np.float32(1.0 / np.sqrt(float(n) * (qmax if per_channel else (qmax - qmin)))))


def test_lsq_backward_equals_reference_kernel_goldens(CUDA, gold):
    """LinearQuantize_T_B / _C_B: grad_x == the reference kernel's mask (values ON the clip edges, all 8 rounding modes,
    fractional offsets), grad_s within 1e-4 relative (+ 1e-5 of the factor-scaled sum |dy|: float summation order)."""
    for key, x, dy, s, o, axis, qmin, qmax, r in cases.lsq_cases():
        if axis is None: gx, gs = CUDA.LinearQuantize_T_B(_dev(x), _dev(s), _dev(o), _dev(dy), qmin, qmax, r)
        else: gx, gs = CUDA.LinearQuantize_C_B(_dev(x), _dev(s), _dev(o), _dev(dy), qmin, qmax, axis, r)
        mask = np.unpackbits(gold[key + '_mask'])[:x.size].astype(bool)
        gx = gx.cpu().numpy().reshape(-1)
        assert np.array_equal(gx != 0, mask), key
        assert np.array_equal(gx[mask], dy.reshape(-1)[mask]), key
        f = _factor(x.size, qmin, qmax, axis is not None)
        want = (gold[key + '_psum'] * f).astype(np.float32).reshape(-1)
        assert np.allclose(gs.cpu().numpy().reshape(-1), want, rtol=1e-4, atol=1e-5 * float(np.abs(dy).sum()) * f), (key, gs, want)


def test_fp8_backward_equals_reference_kernel_goldens(CUDA, gold):
    for key, x, dy, s, o, axis, E, M, c in cases.fp8_bwd_cases():
        if axis is None: gx, gs = CUDA.FloatingQuantize_T_B(_dev(x), _dev(s), _dev(o), _dev(dy), E, M, -c, c, 0)
        else: gx, gs = CUDA.FloatingQuantize_C_B(_dev(x), _dev(s), _dev(o), _dev(dy), E, M, -c, c, axis, 0)
        mask = np.unpackbits(gold[key + '_mask'])[:x.size].astype(bool)
        assert np.array_equal(gx.cpu().numpy().reshape(-1) != 0, mask), key
        assert np.allclose(gs.cpu().numpy().reshape(-1), gold[key + '_gs'].reshape(-1), rtol=5e-4, atol=1e-5), (key, gs, gold[key + '_gs'])


# ------------------------------------------------------------------------------------------- the library itself, larger inputs
@needs_ref
def test_histograms_equal_reference_kernel_source_large(CUDA):
    """1.6 M elements (B = [1,512,56,56]) through the reference's kernel text on the host vs the HIP kernels: randn and ReLU
    data, 2048 and the reference's default 4096 bins (core/common.py:18), sym / asym / per channel."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 512, 56, 56, generator=g) * 1.3
    for name, t in (('randn', x), ('relu', torch.relu(x))):
        v = t.numpy()
        for bins in (2048, 4096):
            hs = float(np.float32(float(t.abs().max()) / bins))
            for clip in (True, False):
                h = torch.zeros(bins, dtype=torch.int32, device=DEV)
                CUDA.Histogram_T(t.to(DEV), h, hs, clip)
                assert np.array_equal(h.cpu().numpy(), K.hist_sym_t(v, hs, np.zeros(bins, np.int32), clip)), (name, bins, clip)
            lo, hi = float(t.min()), float(t.max())
            h = torch.zeros(bins, dtype=torch.int32, device=DEV)
            CUDA.Histogram_Asymmetric_T(lo, hi, t.to(DEV), h, True)
            assert np.array_equal(h.cpu().numpy(), K.hist_asym_t(v, lo, hi, np.zeros(bins, np.int32), True)), (name, bins)
        h = torch.zeros(512, 256, dtype=torch.int32, device=DEV)
        hs = float(np.float32(float(t.abs().max()) / 256))
        CUDA.Histogram_C(t.to(DEV), 1, h, hs, True)
        assert np.array_equal(h.cpu().numpy().reshape(-1), K.hist_sym_c(v, 1, hs, np.zeros(512 * 256, np.int32), True)), name


@needs_ref
def test_lsq_backward_equals_reference_kernel_source_large(CUDA):
    """Weight-sized and activation-sized tensors through `_QuantizeTensor_LT_B / _LC_B` on the host vs HIP: mask exact,
    grad_s 1e-4 relative of the term scale (the reference's own test tolerates an SNR of 1e-3, tests/test_cuda_kernel.py:96-101)."""
    rng = np.random.default_rng(8)
    for shape, axis in (((64, 32, 3, 3), 0), ((2, 24, 28, 28), 1), ((5, 7, 11), 2)):
        x = (rng.standard_normal(shape) * 0.7).astype(np.float32); dy = rng.standard_normal(shape).astype(np.float32)
        C = shape[axis]
        for qmin, qmax in ((-8, 7), (-128, 127)):
            s1 = np.array([np.abs(x).max() / qmax * 0.6], np.float32); o1 = np.zeros(1, np.float32)
            gx, gs = CUDA.LinearQuantize_T_B(_dev(x), _dev(s1), _dev(o1), _dev(dy), qmin, qmax, 0)
            wx, ws = K.fq_linear_t_bwd(x, s1, o1, dy, qmin, qmax, 0)
            assert np.array_equal(gx.cpu().numpy(), wx)
            assert np.allclose(gs.cpu().numpy(), ws, rtol=1e-4, atol=1e-5 * np.abs(dy).sum() * _factor(x.size, qmin, qmax, False))
            sc = (np.abs(x).max() / qmax * rng.uniform(0.3, 1.0, C)).astype(np.float32); oc = rng.integers(-2, 3, C).astype(np.float32)
            gx, gs = CUDA.LinearQuantize_C_B(_dev(x), _dev(sc), _dev(oc), _dev(dy), qmin, qmax, axis, 0)
            wx, ws = K.fq_linear_c_bwd(x, sc, oc, dy, axis, qmin, qmax, 0)
            assert np.array_equal(gx.cpu().numpy(), wx)
            assert np.allclose(gs.cpu().numpy(), ws, rtol=1e-4, atol=1e-5 * np.abs(dy).sum() / C * _factor(x.size, qmin, qmax, True) * 10)
