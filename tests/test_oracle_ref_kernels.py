"""CPU: the oracle's restatements of the histogram / quantile / LSQ-backward / FP8-backward kernels against outputs PRODUCED BY
THE REFERENCE'S OWN KERNEL BODIES (VERDICT r3 "Missing 2").

``tests/golden/kernels_ref.npz`` was written by ``tests/golden/make_golden.py::gen_kernels_ref`` from
``oracle/_ref/libref_kernels.so`` = the ``__global__`` functions of ``ppq/csrc/cuda/{sort,linear,floating}.cu`` extracted at build
time and executed on the host (``oracle/ref_kernels.py``).  The golden comparisons run everywhere; where the library itself is
present (the build container, and the GPU box: it travels with the snapshot) the goldens are re-derived and wider sweeps run."""
import os
import sys

import numpy as np
import pytest

from oracle import ppq_oracle as O
from oracle import ref_kernels as K

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import kernel_ref_cases as cases  # noqa: E402

needs_ref = pytest.mark.skipif(not K.available(), reason='oracle/_ref/libref_kernels.so not built (needs /root/reference at build time)')


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'kernels_ref.npz'))


def _oracle_hist(kind, v, prm):
    if kind == 'sym': return O.hist_sym_t(v, prm[0], np.zeros(cases.BINS, np.int32), prm[1])
    if kind == 'asym': return O.hist_asym_t(v, prm[0], prm[1], np.zeros(cases.BINS, np.int32), prm[2])
    shape, axis, hs, clip = prm
    return O.hist_sym_c(v, axis, hs, np.zeros(shape[axis] * 128, np.int32), clip)


def test_oracle_histograms_equal_reference_kernel_goldens(gold):
    """Bin rule, outlier rule and channel index of sort.cu:75-89, 113-139, 167-185 on boundary vectors: 8 awkward hist_scales x
    {sym, asym} x {clip, clamp}, per-channel on 3 axes, data-derived asymmetric ranges."""
    n = 0
    for key, kind, v, prm in cases.hist_cases():
        want = gold['hist_' + key]
        got = _oracle_hist(kind, v, prm)
        assert np.array_equal(got, want), (key, np.nonzero(got != want)[0][:8])
        n += 1
    assert n >= 40


def oracle_quantile_positions(n: int, q: float):
    """The oracle's index rule (oracle/ppq_oracle.c: oracle_quantile_t), on its own: rn((float)n * q), clipped."""
    f = np.float32
    def pos(qq):
        p = np.rint(f(n) * f(qq))                            # float32 product of the float32-converted n, RNE
        return int(min(max(int(p), 0), n - 1))
    return pos(f(q)), pos(f(1) - f(q))


def test_oracle_quantile_index_rule_equals_reference_kernel(gold):
    """`__float2int_rn(num_of_elements * q)` (sort.cu:13, 17) for n from 1 to 2^31 - 1, incl. n > 2^24 where the int64 -> float
    conversion of n rounds, q = 0 / 1 / .5 (n * q on a tie) / 0.9999 ...; the positions were observed from the kernel itself."""
    qpos = gold['qpos']
    for i, n in enumerate(cases.QUANTILE_NS):
        for j, q in enumerate(cases.QUANTILE_QS):
            assert oracle_quantile_positions(n, q) == tuple(int(a) for a in qpos[i, j]), (n, q)
    # and the full function (sort + pick) on small tensors
    for key, v, q in cases.quantile_arrays():
        assert np.array_equal(O.quantile_t(v, q), gold[key]), key


def _factor(n, qmin, qmax, per_channel):
    return np.float32(1.0 / np.sqrt(float(n) * (qmax if per_channel else (qmax - qmin))))


def test_oracle_lsq_backward_equals_reference_kernel_goldens(gold):
    """linear.cu:235-282, 326-380 through the kernel bodies: the clip mask (-> grad_x) bit for bit on values sitting ON the clip
    edges and rounding ties, all 8 rounding modes, fractional offsets; grad_s against the double sum of the kernel's own
    per-thread terms (1 ulp: same terms, another summation order) and against the kernel's float result (1e-5 of sum |term|)."""
    n = 0
    for key, x, dy, s, o, axis, qmin, qmax, r in cases.lsq_cases():
        if axis is None: gx, gs = O.fq_linear_t_bwd(x, s, o, dy, qmin, qmax, r)
        else: gx, gs = O.fq_linear_c_bwd(x, s, o, dy, axis, qmin, qmax, r)
        mask = np.unpackbits(gold[key + '_mask'])[:x.size].astype(bool)
        assert np.array_equal(gx.reshape(-1) != 0, mask), key
        assert np.array_equal(gx.reshape(-1)[mask], dy.reshape(-1)[mask]), key
        from_terms = (gold[key + '_psum'] * float(_factor(x.size, qmin, qmax, axis is not None))).astype(np.float32)
        if axis in (None, 0):                    # one term per thread: the same float terms, summed in double on both sides
            assert np.allclose(gs, from_terms.reshape(gs.shape), rtol=3e-7, atol=1e-9), (key, gs, from_terms)
        else:                                    # a thread's term is its FLOAT running sum over the outer index (linear.cu:347-372)
            assert np.allclose(gs, from_terms.reshape(gs.shape), rtol=3e-6, atol=3e-6), (key, gs, from_terms)
        assert np.allclose(gs, gold[key + '_gs'], rtol=1e-4, atol=1e-5 * np.abs(dy).sum() * float(_factor(x.size, qmin, qmax, axis is not None))), key
        n += 1
    assert n == 72


def test_oracle_fp8_backward_equals_reference_kernel_goldens(gold):
    """floating.cu:133-182, 223-283: saturation mask (qt == clip +- 1) exact, grad_s to float-summation tolerance."""
    for key, x, dy, s, o, axis, E, M, c in cases.fp8_bwd_cases():
        gx, gs = O.fq_float_c_bwd(x, s, o, dy, axis, E, M, -c, c, 0)
        mask = np.unpackbits(gold[key + '_mask'])[:x.size].astype(bool)
        assert np.array_equal(gx.reshape(-1) != 0, mask), key
        assert np.allclose(gs, gold[key + '_gs'], rtol=2e-4, atol=1e-6), (key, gs, gold[key + '_gs'])


# ------------------------------------------------------------------------------------------------ library present
@needs_ref
def test_goldens_are_what_the_reference_kernels_return_now(gold):
    """Regeneration check: the committed file equals what libref_kernels.so returns here (catches a stale golden file)."""
    for key, kind, v, prm in cases.hist_cases()[::5]:
        if kind == 'sym': got = K.hist_sym_t(v, prm[0], np.zeros(cases.BINS, np.int32), prm[1])
        elif kind == 'asym': got = K.hist_asym_t(v, prm[0], prm[1], np.zeros(cases.BINS, np.int32), prm[2])
        else: got = K.hist_sym_c(v, prm[1], prm[2], np.zeros(prm[0][prm[1]] * 128, np.int32), prm[3])
        assert np.array_equal(got, gold['hist_' + key]), key
    for i, n in list(enumerate(cases.QUANTILE_NS))[::3]:
        for j, q in enumerate(cases.QUANTILE_QS):
            assert K.quantile_positions(n, float(np.float32(q))) == tuple(int(a) for a in gold['qpos'][i, j])


@needs_ref
def test_oracle_equals_reference_kernels_random_sweep():
    """Wider than the goldens: random tensors / scales / ranges / shapes through both, forward kernels included (the
    vectorised and the scalar kernel variants of linear.cu:38-86, 132-186 are both reached: sizes divisible by 4 and not)."""
    rng = np.random.default_rng(2024)
    for it in range(40):
        shape = tuple(int(a) for a in rng.integers(1, 9, size=rng.integers(1, 5)))
        x = (rng.standard_normal(shape) * 10 ** rng.uniform(-3, 3)).astype(np.float32)
        bins = int(rng.choice([64, 2048, 4096]))
        hs = float(np.float32(np.abs(x).max() / bins * rng.uniform(0.3, 1.2) + 1e-30))
        clip = bool(it & 1)
        assert np.array_equal(O.hist_sym_t(x, hs, np.zeros(bins, np.int32), clip), K.hist_sym_t(x, hs, np.zeros(bins, np.int32), clip))
        lo, hi = float(x.min() - rng.uniform(0, 1) * hs), float(x.max() * rng.uniform(0.5, 1.0) + hs)
        if hi > lo:
            assert np.array_equal(O.hist_asym_t(x, lo, hi, np.zeros(bins, np.int32), clip), K.hist_asym_t(x, lo, hi, np.zeros(bins, np.int32), clip))
        axis = int(rng.integers(0, len(shape)))
        C = shape[axis]
        assert np.array_equal(O.hist_sym_c(x, axis, hs, np.zeros(C * 64, np.int32), clip), K.hist_sym_c(x, axis, hs, np.zeros(C * 64, np.int32), clip))
        q = float(rng.choice([0.9999, 0.999, 0.5, 0.01]))
        assert np.array_equal(O.quantile_t(x, q), K.quantile_t(x, q))
        assert np.array_equal(O.isotone_t(x), K.isotone_t(x))
        r = int(rng.integers(0, 8))
        qmin, qmax = [(-128, 127), (0, 255), (-8, 7)][it % 3]
        xs = (x / max(float(np.abs(x).max()), 1e-20) * 40).astype(np.float32)
        s1 = np.array([rng.uniform(0.05, 1.0)], np.float32); o1 = np.array([rng.integers(qmin, qmax) + 0.5 * (it & 1)], np.float32)
        assert np.array_equal(O.fq_linear_t(xs, s1, o1, qmin, qmax, r).view(np.uint32), K.fq_linear_t(xs, s1, o1, qmin, qmax, r).view(np.uint32))
        sc = rng.uniform(0.05, 1.0, C).astype(np.float32); oc = rng.integers(qmin, qmax, C).astype(np.float32)
        assert np.array_equal(O.fq_linear_c(xs, sc, oc, axis, qmin, qmax, r).view(np.uint32), K.fq_linear_c(xs, sc, oc, axis, qmin, qmax, r).view(np.uint32))
        dy = rng.standard_normal(shape).astype(np.float32)
        a, ga = O.fq_linear_t_bwd(xs, s1, o1, dy, qmin, qmax, r); b, gb = K.fq_linear_t_bwd(xs, s1, o1, dy, qmin, qmax, r)
        assert np.array_equal(a, b) and np.allclose(ga, gb, rtol=1e-4, atol=1e-5)
        a, ga = O.fq_linear_c_bwd(xs, sc, oc, dy, axis, qmin, qmax, r); b, gb = K.fq_linear_c_bwd(xs, sc, oc, dy, axis, qmin, qmax, r)
        assert np.array_equal(a, b) and np.allclose(ga, gb, rtol=1e-4, atol=1e-5)
        xf = (xs * rng.choice([0.01, 1, 20])).astype(np.float32)
        sf = np.array([2.0 ** int(rng.integers(-4, 3))], np.float32); z = np.zeros(1, np.float32)
        assert np.array_equal(O.fq_float_t(xf, sf, z).view(np.uint32), K.fq_float_t(xf, sf, z).view(np.uint32))
        scf = (2.0 ** rng.integers(-4, 3, C)).astype(np.float32)
        assert np.array_equal(O.fq_float_c(xf, scf, np.zeros(C, np.float32), axis).view(np.uint32),
                              K.fq_float_c(xf, scf, np.zeros(C, np.float32), axis).view(np.uint32))
        a, ga = O.fq_float_c_bwd(xf, scf, np.zeros(C, np.float32), dy, axis, 4, 3, -448.0, 448.0)
        b, gb = K.fq_float_c_bwd(xf, scf, np.zeros(C, np.float32), dy, axis, 4, 3, -448.0, 448.0)
        assert np.array_equal(a, b) and np.allclose(ga, gb, rtol=2e-4, atol=1e-5)
