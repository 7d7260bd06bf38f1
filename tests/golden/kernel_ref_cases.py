"""Cases + seeded inputs of tests/golden/kernels_ref.npz: imported by make_golden.py (which runs the REFERENCE'S OWN kernel
bodies on them through oracle/_ref/libref_kernels.so) and by the tests (which run the oracle and the HIP kernels on them).
numpy only, no reference import: this file travels to the GPU box.

Inputs stay finite with |x| / hist_scale < 2^31: the reference's `int b = floor(...)` is undefined on a host outside int32
(oracle/ref_kernel_host.h); the device-side saturation behaviour is covered by tests/test_gpu_kernels.py against the oracle."""
import numpy as np

BINS = 2048
HIST_SCALES = [0.01, 1.0 / 3.0, 0.1, 0.007812501, 1e-3, 3.3333333e-5, 7.0, 1e-30]


def boundary_values(hs: float, seed: int, n: int = 4000) -> np.ndarray:
    """Values exactly on / one ulp around the bin boundaries k * hs, a near-integer-quotient family, uniform filler,
    zeros of both signs and a denormal: the vectors of test_hist_boundary_exactness, minus inf / NaN / 3e38."""
    rng = np.random.default_rng(seed)
    hs32 = np.float32(hs)
    k = rng.integers(0, BINS + 40, size=n).astype(np.float32)
    edge = (k * hs32).astype(np.float32)
    vals = np.concatenate([edge, np.nextafter(edge, np.float32(np.inf)), np.nextafter(edge, np.float32(-np.inf)),
                           (edge * np.float32(1 + 2 ** -22)).astype(np.float32),
                           (rng.random(n) * 2100 * float(hs32)).astype(np.float32),
                           np.array([0.0, -0.0, 1e-45, float(hs32) * 2047.9999, float(hs32) * 2048.0], np.float32)])
    return (vals * rng.choice([-1.0, 1.0], size=vals.size)).astype(np.float32)


def hist_cases():
    """(key, kind, values, params): kind 'sym' -> (hist_scale, clip), 'asym' -> (min, max, clip), 'sym_c' -> (shape, axis, hist_scale, clip)."""
    out = []
    for i, hs in enumerate(HIST_SCALES):
        v = boundary_values(hs, 100 + i)
        hs32 = float(np.float32(hs))
        for clip in (True, False):
            out.append((f'sym_{i}_{int(clip)}', 'sym', v, (hs32, clip)))
        lo = float(np.float32(-hs32 * 1000))
        hi = float(np.float32(lo + hs32 * BINS))
        for clip in (True, False):
            out.append((f'asym_{i}_{int(clip)}', 'asym', v, (lo, hi, clip)))
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((3, 8, 5, 6)) * 2).astype(np.float32)
    x.reshape(-1)[::9] = (np.round(x.reshape(-1)[::9] / 0.02) * np.float32(0.02)).astype(np.float32)     # on boundaries of hs = 0.02
    for axis in (0, 1, 3):
        for clip in (True, False):
            out.append((f'symc_{axis}_{int(clip)}', 'sym_c', x, (x.shape, axis, float(np.float32(0.02)), clip)))
    # asymmetric ranges that do not start at a multiple of the bin width, and an activation-like range [min, max] of the data
    y = (rng.standard_normal(30000) * 1.7 + 0.4).astype(np.float32)
    out.append(('asym_data_1', 'asym', y, (float(y.min()), float(y.max()), True)))
    out.append(('asym_data_0', 'asym', y, (float(y.min()), float(y.max()), False)))
    out.append(('asym_narrow_1', 'asym', y, (-0.73, 1.91, True)))
    out.append(('asym_narrow_0', 'asym', y, (-0.73, 1.91, False)))
    return out


QUANTILE_QS = [0.9999, 0.99999, 0.999, 0.99, 0.5, 0.0, 1.0, 1e-4, 0.75]
QUANTILE_NS = [1, 2, 3, 5, 10, 1000, 1024, 1025, 4999, 5000, 5001, 15000, 150528, 1605632, 16777215, 16777216, 16777217, 16777219,
               33554433, 51380224, 205520896, 537181440, 1073754169, 2147483647]


def quantile_arrays():
    """Small tensors whose (max, min) pair the reference kernel picks after the sort: (key, values, q)."""
    rng = np.random.default_rng(3)
    out = []
    for n in (1, 2, 3, 7, 1000, 5001, 40000):
        v = (rng.standard_normal(n) * 3).astype(np.float32)
        if n >= 1000: v[::5] = np.maximum(v[::5], 0)              # ReLU-like ties at 0
        for q in (0.9999, 0.99, 0.5, 0.0, 1.0):
            out.append((f'q_{n}_{q}', v, q))
    return out


def lsq_cases():
    """(key, x, dy, scale, offset, channel_axis or None, qmin, qmax, rounding).  x / scale lands ON the clip edges
    (qmin - .5, qmin, qmax, qmax + .5 ...) and on rounding ties; fractional offsets (rounded half away in the kernel)."""
    rng = np.random.default_rng(11)
    out = []
    for r in range(8):
        for (qmin, qmax, off) in ((-8, 7, 0.0), (0, 255, 127.5), (-128, 127, -2.5)):
            shape = (6, 5, 7, 3)
            s = np.float32(0.0625)
            x = (rng.standard_normal(shape) * (qmax - qmin) * 0.4 * s).astype(np.float32)
            flat = x.reshape(-1)
            o = np.float32(np.round(off)) if off >= 0 else np.float32(-np.round(-off))
            edges = np.array([qmin - 1, qmin - 0.5, qmin, qmin + 0.5, qmax - 0.5, qmax, qmax + 0.5, qmax + 1, 0.5, -0.5, 1.5, 2.5], np.float32)
            flat[:edges.size * 3:3] = ((edges - o) * s).astype(np.float32)
            dy = rng.standard_normal(shape).astype(np.float32)
            dy[dy == 0] = 1.0
            out.append((f'lt_r{r}_{qmin}', x, dy, np.array([s], np.float32), np.array([off], np.float32), None, qmin, qmax, r))
            for axis in (0, 1):
                C = shape[axis]
                sc = (2.0 ** rng.integers(-5, -2, C)).astype(np.float32)
                sc[1] = np.float32(0.0371)
                oc = (np.full(C, off) + rng.integers(-2, 3, C)).astype(np.float32)
                out.append((f'lc_r{r}_{qmin}_a{axis}', x, dy, sc, oc, axis, qmin, qmax, r))
    return out


def fp8_bwd_cases():
    """(key, x, dy, scale, offset, channel_axis or None, exponent, mantissa, clip) for _QuantizeTensor_FT_B / _FC_B."""
    rng = np.random.default_rng(13)
    out = []
    for name, (E, M, c) in (('e4m3', (4, 3, 448.0)), ('e5m2', (5, 2, 57344.0))):
        shape = (4, 6, 50)
        x = (rng.standard_normal(shape) * rng.choice([1e-3, 0.1, 3, 90, c, 3 * c], shape)).astype(np.float32)
        x.reshape(-1)[:6] = np.array([c, -c, c * 1.0001, -c * 1.0001, 0.0, 2.0 ** -9], np.float32)
        dy = rng.standard_normal(shape).astype(np.float32)
        dy[dy == 0] = 1.0
        out.append((f'ft_{name}', x, dy, np.array([1.0], np.float32), np.array([0.0], np.float32), None, E, M, c))
        out.append((f'ft_{name}_s', x, dy, np.array([0.25], np.float32), np.array([0.0], np.float32), None, E, M, c))
        for axis in (0, 1):
            C = shape[axis]
            s = (2.0 ** rng.integers(-4, 3, C)).astype(np.float32)
            out.append((f'fc_{name}_a{axis}', x, dy, s, np.zeros(C, np.float32), axis, E, M, c))
    return out
