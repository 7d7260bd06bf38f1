"""Generates tests/golden/kl_near_tie.npz: ONE 2048-bin histogram (ReLU-shaped data, numpy default_rng(28)) on which the KL search
of observer/range.py:190-282 has two NEIGHBOURING candidates (bin ranges 1792 and 1920) whose divergences differ by 1.45e-4
relative -- close enough that the float32 summation ORDER of the candidate distribution's normaliser decides the arg-min
(tests/test_oracle_golden.py::test_kl_near_tie_follows_the_float32_sum_order).  Found by scanning 300 seeds with /tmp-only
tooling; the histogram itself is committed so the test does not depend on numpy's random stream.
    python tests/golden/make_kl_near_tie.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ppq_oracle as O  # noqa: E402

seed = 28
r = np.random.default_rng(seed)
n = r.integers(20000, 400000)
x = (np.maximum(r.standard_normal(n), 0) * 3).astype(np.float32)
hs = float(x.max()) / 2048
hist = np.zeros(2048, np.int32)
O.hist_sym_t(x, hs, hist)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'kl_near_tie.npz'), hist=hist, hist_scale=np.float64(hs), seed=seed, n=n)
print('wrote kl_near_tie.npz', int(hist.sum()), n)
