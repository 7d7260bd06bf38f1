"""CPU-side tests (`-m "not gpu"`): the C-ABI library loads and exports every symbol the header
declares, the host arithmetic that mirrors the reference matches the golden vectors, the harness /
pass plumbing is wired like the reference's, and the data-parallel merge is correct over gloo with
world_size 2.  No kernel is launched here (there is no GPU in the build container)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ppq_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'ppq_hip.h')).read()
    declared = set(re.findall(r'\b(ppqhip_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(_lib.lib, name), f'{name} declared in include/ppq_hip.h but not exported'
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert _lib.lib.ppqhip_version() == _lib.ABI_VERSION == 4


def test_no_cpu_fallback_and_error_convention():
    from ppq_amd import CUDA, qfunction, LinearQuantizationConfig, QuantizationStates
    t = torch.zeros(8)
    with pytest.raises(RuntimeError, match='not on the GPU'):
        CUDA.LinearQuantize_T(t, torch.ones(1), torch.zeros(1))
    with pytest.raises(RuntimeError, match='Invalid dtype'):
        CUDA.LinearQuantize_T(t.double(), torch.ones(1), torch.zeros(1))
    with pytest.raises(RuntimeError, match='empty'):
        CUDA.Histogram_T(torch.zeros(0), torch.zeros(4, dtype=torch.int32), 0.1)
    from ppq_amd.ffi import FloatingQuantizePlan, LinearQuantizePlan
    for call in (lambda: CUDA.FloatScaleSearch([(torch.zeros(2, 8), 4, 3, -448., 448.)], [1.0]),
                 lambda: FloatingQuantizePlan([(torch.zeros(2, 8), torch.ones(2), torch.zeros(2), 0, 4, 3, -448., 448.)]),
                 lambda: LinearQuantizePlan([(torch.zeros(2, 8), torch.ones(2), torch.zeros(2), 0, -128, 127)])):
        with pytest.raises(RuntimeError, match='not on the GPU'): call()        # the multi-tensor entry points refuse host tensors too
    cfg = LinearQuantizationConfig()
    assert qfunction.PPQuantFunction(t, cfg) is t              # INITIAL state: untouched (quant.py:358-359)
    cfg.state = QuantizationStates.ACTIVATED; cfg.scale = torch.ones(1); cfg.offset = torch.zeros(1)
    with pytest.raises(RuntimeError, match='not on the GPU'):
        qfunction.PPQuantFunction(t, cfg)
    # the host helper of the reference's extension is a host function: it runs anywhere
    from oracle import ppq_oracle as O
    h = np.random.default_rng(0).integers(0, 1000, 2048)
    assert CUDA.compute_mse_loss(h.tolist(), 8, 2, 520) == O.mse_loss(h, 8, 2, 520)
    if O.ref_lib() is not None:
        assert CUDA.compute_mse_loss(h.tolist(), 8, 2, 520) == O.ref_mse_loss(h, 8, 2, 520)


def test_rounding_mirror_matches_reference_tables(golden_dir):
    from ppq_amd.core import RoundingPolicy as R
    from ppq_amd.round import ppq_numerical_round, ppq_round_to_power_of_2, ppq_tensor_round
    # /root/reference/tests/test_rounding.py:5-37
    assert ppq_numerical_round(1.5, R.ROUND_HALF_EVEN) == 2 and ppq_numerical_round(2.5, R.ROUND_HALF_EVEN) == 2
    assert ppq_numerical_round(0.5, R.ROUND_HALF_EVEN) == 0 and ppq_numerical_round(-0.5, R.ROUND_HALF_EVEN) == 0
    assert [ppq_numerical_round(v, R.ROUND_HALF_UP) for v in (1.5, 2.5, 0.5, -0.5)] == [2, 3, 1, 0]
    assert [ppq_numerical_round(v, R.ROUND_HALF_DOWN) for v in (1.5, 2.5, 0.5, -0.5)] == [1, 2, 0, -1]
    assert [ppq_numerical_round(v, R.ROUND_HALF_TOWARDS_ZERO) for v in (1.5, 2.5, 0.5)] == [1, 2, 0]
    assert ppq_round_to_power_of_2(1.0) == 1 and ppq_round_to_power_of_2(1.2) == 2 and ppq_round_to_power_of_2(3.2) == 4
    assert ppq_round_to_power_of_2(0.26) == 0.5 and ppq_round_to_power_of_2(0.24) == 0.25
    z = np.load(os.path.join(golden_dir, 'rounding.npz'))
    for pol in R:
        assert [ppq_numerical_round(float(v), pol) for v in z['num_values']] == list(z[f'num_{pol.value}'])
        if pol != R.ROUND_TO_NEAR_INT:
            got = ppq_tensor_round(torch.from_numpy(z['grid']), pol).numpy()
            assert np.array_equal(got, z[f'tensor_{pol.value}'])
    assert [ppq_round_to_power_of_2(float(v), R.ROUND_UP) for v in z['pow2_values']] == list(z['pow2_up'])
    assert [ppq_round_to_power_of_2(float(v), R.ROUND_HALF_UP) for v in z['pow2_values']] == list(z['pow2_half_up'])


def test_minmax_to_scale_offset_matches_reference(golden_dir):
    """The product's host arithmetic (ppq_amd.observer.minmax_to_scale_offset) against the scales the
    reference's observers rendered (tests/golden/make_golden.py), fed with oracle ranges."""
    from oracle import ppq_oracle as O
    from ppq_amd import LinearQuantizationConfig
    from ppq_amd.observer import minmax_to_scale_offset
    z = np.load(os.path.join(golden_dir, 'observers.npz'))
    for k in range(int(z['minmax_n'])):
        relu, per_channel, sym, qmin, qmax, bits, pow2 = [int(v) for v in z[f'minmax_{k}_meta']]
        data = np.maximum(z['batches'], 0) if relu else z['batches']
        cfg = LinearQuantizationConfig(symmetrical=bool(sym), power_of_2=bool(pow2), quant_min=qmin, quant_max=qmax,
                                       num_of_bits=bits, channel_axis=1 if per_channel else None)
        if not per_channel:
            mm = None
            for b in data: mm = O.minmax_t(b, mm)
            s, o = minmax_to_scale_offset(float(mm[0]), float(mm[1]), cfg)
            assert np.float32(s) == z[f'minmax_{k}_scale'] and np.float32(o) == z[f'minmax_{k}_offset']
        else:
            mins = maxs = None
            for b in data: mins, maxs = O.minmax_c(b, 1, mins, maxs)
            so = [minmax_to_scale_offset(a, b, cfg) for a, b in zip(mins, maxs)]      # numpy float32 scalars
            assert np.array_equal(np.array([v[0] for v in so], np.float32), z[f'minmax_{k}_scale'])
            assert np.array_equal(np.array([v[1] for v in so], np.float32), z[f'minmax_{k}_offset'])


def test_core_types_mirror_reference_values():
    from ppq_amd import (FloatingQuantizationConfig, LinearQuantizationConfig, QuantizationPolicy,
                         QuantizationProperty as P, QuantizationStates as S, RoundingPolicy as R)
    assert [r.value for r in R] == [0, 1, 2, 3, 4, 5, 6]
    assert (P.PER_TENSOR.value, P.PER_CHANNEL.value, P.LINEAR.value, P.FLOATING.value, P.SYMMETRICAL.value,
            P.ASYMMETRICAL.value, P.POWER_OF_2.value, P.DYNAMIC.value) == (1, 2, 4, 8, 16, 32, 64, 128)
    assert (S.INITIAL.value, S.ACTIVATED.value, S.PASSIVE.value, S.FP32.value) == (1, 4, 5, 8)
    assert S.is_activated(S.ACTIVATED) and S.is_activated(S.PASSIVE) and not S.is_activated(S.INITIAL)
    with pytest.raises(ValueError): QuantizationPolicy(P.LINEAR.value)                       # no granularity
    with pytest.raises(ValueError): QuantizationPolicy(P.FLOATING + P.SYMMETRICAL + P.PER_TENSOR)   # needs POWER_OF_2
    c = LinearQuantizationConfig(symmetrical=False, channel_axis=1, quant_min=0, quant_max=255)
    assert c.policy.has_property(P.PER_CHANNEL) and c.policy.has_property(P.ASYMMETRICAL) and c.channel_axis == 1
    f = FloatingQuantizationConfig()
    assert (f.exponent_bits, f.mantissa_bits, f.quant_min, f.quant_max) == (4, 3, -448.0, 448.0)
    if os.path.isdir('/root/reference'):      # the values are the reference's own
        src = open('/root/reference/ppq/core/quant.py').read()
        for name, val in (('ROUND_HALF_EVEN', 0), ('ROUND_UP', 6)):
            assert re.search(rf'{name}\s*=\s*{val}\b', src)
        assert re.search(r'POWER_OF_2\s*=\s*0x00000040', src) and re.search(r'ACTIVATED\s*=\s*4', src)


def test_cuda_facade_signatures_match_reference():
    """Every static method of ppq.core.ffi.CUDA exists on ppq_amd.CUDA with the same parameter names,
    order and defaults (parsed from the reference source; no import of ppq needed)."""
    if not os.path.isdir('/root/reference'):
        pytest.skip('reference not present on this machine')
    import ast
    import inspect
    from ppq_amd import CUDA
    tree = ast.parse(open('/root/reference/ppq/core/ffi.py').read())
    ref = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'CUDA')
    checked = 0
    for fn in ref.body:
        if not isinstance(fn, ast.FunctionDef) or fn.name == 'OrderPreservingObserve':   # mis-wired in the reference
            continue
        ours = inspect.signature(getattr(CUDA, fn.name))
        ref_args = [a.arg for a in fn.args.args]
        assert list(ours.parameters) == ref_args, (fn.name, list(ours.parameters), ref_args)
        ref_defaults = [ast.literal_eval(ast.unparse(d).replace('- ', '-').replace('+ ', '+')) for d in fn.args.defaults]
        our_defaults = [p.default for p in ours.parameters.values() if p.default is not inspect._empty]
        assert our_defaults == ref_defaults, (fn.name, our_defaults, ref_defaults)
        checked += 1
    assert checked >= 20
    # and the extension object carries the 20 pybind names of csrc/export.cc
    from ppq_amd import HIP_EXTENSION
    names = re.findall(r'm\.def\("(\w+)"', open('/root/reference/ppq/csrc/export.cc').read())
    assert len(set(names)) == 20
    for n in names: assert callable(getattr(HIP_EXTENSION, n)), n


def test_install_into_reference_ppq_routes_calls():
    """Drop-in seam: after install_into_ppq() an UNMODIFIED ppq.core.ffi.CUDA wrapper calls our
    extension object with the pybind argument order (recorded with a spy; no kernel runs)."""
    if not os.path.isdir('/root/reference'):
        pytest.skip('reference not present on this machine')
    import subprocess
    import sys
    code = r'''
import importlib.machinery, os, sys
from unittest.mock import MagicMock
os.environ['PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION'] = 'python'
for n in ['onnx','onnx.helper','onnx.numpy_helper','onnx.mapping','onnx.onnx_pb','onnx.checker','onnx.external_data_helper','onnx.shape_inference','onnx.version_converter']:
    m = MagicMock(); m.__spec__ = importlib.machinery.ModuleSpec(n, None); m.__path__ = []; sys.modules[n] = m
sys.path.insert(0, '/root/reference'); sys.path.insert(0, %r)
import torch, ppq
import ppq_amd
from ppq.core import CUDA, PPQ_CONFIG
ppq_amd.install_into_ppq()
assert PPQ_CONFIG.USING_CUDA_KERNEL is True
calls = []
ext = ppq_amd.HIP_EXTENSION
for name in ['QuantizeTensor_LT', 'QuantizeTensor_LC', 'QuantizeTensor_LC_B', 'Histogram_T', 'Histogram_Asymmetric_T', 'QuantizeTensor_FC', 'Quantile_T']:
    setattr(type(ext), name, staticmethod((lambda nm: lambda *a: calls.append((nm, a)) or a[0])(name)))
t = torch.zeros(2, 3); s = torch.ones(3); o = torch.zeros(3); h = torch.zeros(4, dtype=torch.int32)
CUDA.LinearQuantize_T(t, s, o, -8, 7, 0)
CUDA.LinearQuantize_C(t, s, o, 1, -8, 7, 0)
CUDA.LinearQuantize_C_B(t, s, o, t, -8, 7, 1, 0)
CUDA.Histogram_T(t, h, 0.5, False)
CUDA.Histogram_Asymmetric_T(-1.0, 1.0, t, h, True)
CUDA.FloatingQuantize_C(t, s, o, 1, 5, 2, -57344.0, 57344.0, 0)
CUDA.Quantile(t, 0.99)
assert calls[0] == ('QuantizeTensor_LT', (t, s, o, -8, 7, 0))
assert calls[1][0] == 'QuantizeTensor_LC' and calls[1][1][3:] == (-8, 7, 1, 0)          # min, max, axis, rounding
assert calls[2][0] == 'QuantizeTensor_LC_B' and calls[2][1][4:] == (-8, 7, 0, 1)        # min, max, rounding, axis
assert calls[3][0] == 'Histogram_T' and calls[3][1][1:3] == (0.5, False) and calls[3][1][3] is h
assert calls[4][0] == 'Histogram_Asymmetric_T' and calls[4][1][:2] == (-1.0, 1.0) and calls[4][1][3] is True
assert calls[5][0] == 'QuantizeTensor_FC' and calls[5][1][3:] == (5, 2, -57344.0, 57344.0, 1, 0)
assert calls[6] == ('Quantile_T', (t, 0.99))
# the idiom of scripts written for the reference: ENABLE_CUDA_KERNEL() calls CUDA_COMPLIER.complie() (api/interface.py:925-928),
# which would JIT-build ppq/csrc -- after install it re-selects this library instead, and uninstall gives the method back
from ppq.api.interface import ENABLE_CUDA_KERNEL
from ppq.core.ffi import CUDA_COMPLIER, ComplieHelper
PPQ_CONFIG.USING_CUDA_KERNEL = False
with ENABLE_CUDA_KERNEL():
    assert PPQ_CONFIG.USING_CUDA_KERNEL is True and CUDA_COMPLIER.CUDA_EXTENSION is ext
    CUDA.LinearQuantize_T(t, s, o, -8, 7, 0)
assert PPQ_CONFIG.USING_CUDA_KERNEL is False and len(calls) == 8
ppq_amd.uninstall_from_ppq()
assert 'complie' not in vars(CUDA_COMPLIER) and CUDA_COMPLIER.complie.__func__ is ComplieHelper.complie
with ppq_amd.ENABLE_CUDA_KERNEL(): assert ppq_amd.PPQ_CONFIG.USING_CUDA_KERNEL is True
print('ROUTED', len(calls))
''' % ROOT
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert 'ROUTED 8' in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_harness_graph_and_pass_plumbing():
    from ppq_amd import harness
    from ppq_amd.calibration import QuantizationOptimizationPass, RuntimeCalibrationPass
    g = harness.resnet50_graph(seed=0)
    assert sum(1 for op in g.operations.values() if op.type == 'Conv') == 53
    assert sum(1 for op in g.operations.values() if op.type == 'Gemm') == 1
    harness.quantize_graph(g, 'kl', hist_bins=2048)
    acts = [(c, v) for op in g.operations.values() for c, v in op.config_with_variable
            if not v.is_parameter and c.state.value == 1]
    assert len(acts) == 72 and all(c.observer_algorithm == 'kl' for c, _ in acts)
    assert all(c.detail['OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE'] == 2048 for c, _ in acts)
    weights = [c for op in g.operations.values() for c, v in op.config_with_variable if v.is_parameter and c.state.value == 1]
    assert len(weights) == 54 and all(c.channel_axis == 0 and c.observer_algorithm == 'minmax' for c in weights)
    p = RuntimeCalibrationPass(method='kl')
    assert isinstance(p, QuantizationOptimizationPass) and p.name == 'PPQ Runtime Calibration Pass'
    with pytest.raises(AssertionError, match='Insufficient Calibration'):
        p.optimize(g, dataloader=[], executor=None, calib_steps=4)
    with pytest.raises(AssertionError, match='too large'):
        p.optimize(g, dataloader=[], executor=None, calib_steps=1024)


def _merge_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ppq_amd.distributed import merge_observers, shard_batches

    class FakeObserver:
        def __init__(self, bufs): self.bufs = bufs
        def reducible(self): return self.bufs

    class FakeRowSet:                       # the isotone observer's protocol: rows are gathered, not reduced
        def __init__(self, rows): self.rows, self.got = rows, None
        def reducible(self): return []
        def gatherable(self): return [self.rows]
        def take_gathered(self, merged): self.got = merged[0]

    g = torch.Generator().manual_seed(100 + rank)
    rng1 = torch.tensor([float(torch.randn(1, generator=g)) - 1, float(torch.randn(1, generator=g)) + 1])
    chan = torch.stack([torch.randn(5, generator=g) - 1, torch.randn(5, generator=g) + 1])
    hist = torch.randint(0, 1000, [64], generator=g, dtype=torch.int32)
    pct = torch.tensor([1.5 * (rank + 1), -2.0 * (rank + 1), 4.0])
    sse = torch.tensor([0.25 * (rank + 1), 1.0], dtype=torch.float64)          # FP8 'floating': per-candidate squared errors
    pairs = FakeRowSet(torch.tensor([[10.0 * rank + i, float(i)] for i in range(rank + 2)]))     # 2 rows on rank 0, 3 on rank 1
    obs = [FakeObserver([(rng1[0:1], 'min'), (rng1[1:2], 'max')]), FakeObserver([(chan[0], 'min'), (chan[1], 'max')]),
           FakeObserver([(hist, 'sum')]), FakeObserver([(pct, 'sum')]), FakeObserver([]), FakeObserver([(sse, 'sum')]), pairs]
    issued = merge_observers(obs)
    shard = shard_batches(list(range(10)))
    q.put((rank, issued, rng1.tolist(), chan.tolist(), hist.tolist(), pct.tolist(), shard, sse.tolist(), pairs.got.tolist()))
    dist.destroy_process_group()


def test_merge_observers_gloo_world2():
    """One MIN all-reduce for all ranges, one SUM per dtype for histograms / percentile sums; the
    merged statistics equal the single-process reduction over both shards and agree on all ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_merge_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs: p.join(timeout=60)
    # expected, recomputed locally
    exp = []
    for rank in range(2):
        g = torch.Generator().manual_seed(100 + rank)
        rng1 = torch.tensor([float(torch.randn(1, generator=g)) - 1, float(torch.randn(1, generator=g)) + 1])
        chan = torch.stack([torch.randn(5, generator=g) - 1, torch.randn(5, generator=g) + 1])
        hist = torch.randint(0, 1000, [64], generator=g, dtype=torch.int32)
        exp.append((rng1, chan, hist))
    want_rng = [min(exp[0][0][0], exp[1][0][0]).item(), max(exp[0][0][1], exp[1][0][1]).item()]
    want_chan = [torch.minimum(exp[0][1][0], exp[1][1][0]).tolist(), torch.maximum(exp[0][1][1], exp[1][1][1]).tolist()]
    want_hist = (exp[0][2] + exp[1][2]).tolist()
    for rank, issued, rng1, chan, hist, pct, shard, sse, pairs in res:
        assert issued == 6                                   # MIN(float) + SUM(int32, float32, float64) + 2 all-gathers (row counts, rows)
        assert rng1 == want_rng and chan == want_chan and hist == want_hist
        assert pct == [4.5, -6.0, 8.0]
        assert sse == [0.75, 2.0]
        assert pairs == [[0.0, 0.0], [1.0, 1.0], [10.0, 0.0], [11.0, 1.0], [12.0, 2.0]]      # rank order, ragged row counts
        assert shard == list(range(rank, 10, 2))


def _overflow_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ppq_amd import distributed as D

    class FakeObserver:
        def __init__(self, bufs): self.bufs = bufs
        def reducible(self): return self.bufs
    big = torch.tensor([1_200_000_000, 5, 7 + rank], dtype=torch.int32)     # 2 x 1.2e9 = 2.4e9 > 2^31 - 1 in bin 0
    try:
        D.merge_observers([FakeObserver([(big, 'sum')])]); first = 'no error'
    except OverflowError as e:
        first = str(e)
    fits = torch.tensor([1_500_000_000 if rank == 0 else 500_000_000, 5, 7 + rank], dtype=torch.int32)    # 2 x 1.5e9 could wrap: widened; the real sum 2e9 fits
    D.merge_observers([FakeObserver([(fits, 'sum')])])
    q.put((rank, first, fits.tolist(), D.last_merge_stats.get('sum_int32_widened', 0)))
    dist.destroy_process_group()


def test_merge_guards_the_int32_histogram_limit():
    """The reference's histograms are int32 (sort.cu:91-165); summed over ranks a bin can pass 2^31.  The layout probe carries
    the largest per-rank count, so the common case stays one int32 SUM; when world_size x that count could wrap, the merge sums
    in int64 and raises on a real overflow instead of wrapping silently."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_overflow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs: p.join(timeout=60)
    for rank, first, fits, widened in res:
        assert 'reaches 2^31' in first, first
        assert fits == [2_000_000_000, 10, 15] and widened == 1


def _world8_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ppq_amd import distributed as D

    class Ob:
        def __init__(self, bufs): self.bufs = bufs
        def reducible(self): return self.bufs

    class RowSet:
        def __init__(self, rows): self.rows, self.got = rows, None
        def reducible(self): return []
        def gatherable(self): return [self.rows]
        def take_gathered(self, merged): self.got = merged[0]
    g = torch.Generator().manual_seed(500 + rank)
    # (1) the ordinary two-phase payload at ResNet-50 size: 72 ranges, 72 x 2048 int32 histograms, percentile sums, FP8 SSEs, row sets
    rngs = [torch.stack([-torch.rand(1, generator=g), torch.rand(1, generator=g)]).reshape(2) for _ in range(72)]
    chan = torch.stack([torch.randn(64, generator=g) - 1, torch.randn(64, generator=g) + 1])
    hists = [torch.randint(0, 1000, [2048], generator=g, dtype=torch.int32) for _ in range(72)]
    pct = torch.tensor([1.0 + rank, -2.0 - rank, 1.0])
    sse = torch.arange(8, dtype=torch.float64) * (rank + 1)
    rows = RowSet(torch.tensor([[100.0 * rank + i, float(i)] for i in range((rank * 3) % 5)]).reshape(-1, 2))     # ragged: 0,3,1,4,2,0,3,1 rows
    obs = [Ob([(r[0:1], 'min'), (r[1:2], 'max')]) for r in rngs] + [Ob([(chan[0], 'min'), (chan[1], 'max')])] + \
          [Ob([(h, 'sum')]) for h in hists] + [Ob([(pct, 'sum')]), Ob([(sse, 'sum')]), rows]
    issued = D.merge_observers(obs)
    stats1 = dict(D.last_merge_stats)
    digest = (float(sum(float(r[0]) for r in rngs)), float(sum(float(r[1]) for r in rngs)), float(chan[0].sum()), float(chan[1].sum()),
              int(sum(int(h.sum()) for h in hists)), pct.tolist(), sse.tolist(), rows.got.tolist())
    # (2) forced widening: 8 x 3e8 = 2.4e9 could wrap -> int64 sum; the true sums fit (bin 0) ...
    fits = torch.tensor([300_000_000 if rank == 0 else 100_000_000, rank, 1], dtype=torch.int32)
    D.merge_observers([Ob([(fits, 'sum')])])
    widened = D.last_merge_stats.get('sum_int32_widened', 0)
    # ... and a real overflow raises on every rank
    big = torch.tensor([300_000_000, 1], dtype=torch.int32)
    try: D.merge_observers([Ob([(big, 'sum')])]); overflow = 'no error'
    except OverflowError as e: overflow = str(e)
    # (3) one rank (5) lacks an observer: the documented error on EVERY rank, no hang
    obs3 = [Ob([(torch.zeros(16, dtype=torch.int32), 'sum')])]
    if rank != 5: obs3.append(Ob([(torch.zeros(16, dtype=torch.int32), 'sum')]))
    try: D.merge_observers(obs3); mismatch = 'no error'
    except RuntimeError as e: mismatch = str(e)
    q.put((rank, issued, stats1.get('world_size'), digest, fits.tolist(), widened, overflow, mismatch))
    dist.destroy_process_group()


def test_merge_observers_gloo_world8():
    """The merge at the world size the north star names (8 ranks): layout probe, one MIN + one SUM per dtype +
    the ragged row-set gather (two ranks contribute ZERO rows), the forced int32 -> int64 widening, a real overflow, and a rank with
    a missing observer -- the merged statistics equal the single-process reduction over the eight shards on every rank."""
    import torch.multiprocessing as mp
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs: p.join(timeout=60)
    mins = maxs = 0.0; cmin = cmax = None; total = 0; want_rows = []
    lo_sum, hi_sum = torch.zeros(72), torch.zeros(72)
    per = []
    for rank in range(world):
        g = torch.Generator().manual_seed(500 + rank)
        rngs = [torch.stack([-torch.rand(1, generator=g), torch.rand(1, generator=g)]).reshape(2) for _ in range(72)]
        chan = torch.stack([torch.randn(64, generator=g) - 1, torch.randn(64, generator=g) + 1])
        hists = [torch.randint(0, 1000, [2048], generator=g, dtype=torch.int32) for _ in range(72)]
        per.append((torch.stack(rngs), chan, sum(int(h.sum()) for h in hists)))
        want_rows += [[100.0 * rank + i, float(i)] for i in range((rank * 3) % 5)]
    all_r = torch.stack([p_[0] for p_ in per])                      # [world, 72, 2]
    want_lo = float(sum(float(v) for v in all_r[:, :, 0].min(0).values)); want_hi = float(sum(float(v) for v in all_r[:, :, 1].max(0).values))
    want_cmin = float(torch.stack([p_[1][0] for p_ in per]).min(0).values.sum()); want_cmax = float(torch.stack([p_[1][1] for p_ in per]).max(0).values.sum())
    want_total = sum(p_[2] for p_ in per)
    for rank, issued, ws, digest, fits, widened, overflow, mismatch in res:
        assert issued == 6 and ws == 8
        lo, hi, cmn, cmx, tot, pct, sse, rows = digest
        assert abs(lo - want_lo) < 1e-4 and abs(hi - want_hi) < 1e-4 and abs(cmn - want_cmin) < 1e-3 and abs(cmx - want_cmax) < 1e-3
        assert tot == want_total
        assert pct == [float(sum(1 + r for r in range(8))), float(sum(-2 - r for r in range(8))), 8.0]
        assert sse == [float(k * 36) for k in range(8)]
        assert rows == want_rows
        assert fits == [1_000_000_000, 28, 8] and widened == 1
        assert 'reaches 2^31' in overflow, overflow
        assert 'different statistics layouts' in mismatch, mismatch


def test_merge_is_noop_without_process_group():
    from ppq_amd.distributed import merge_observers, shard_batches

    class Ob:
        def reducible(self): raise AssertionError('must not be touched when not distributed')
    assert merge_observers([Ob()]) == 0
    assert shard_batches([1, 2, 3]) == [1, 2, 3]


def test_observation_queue_grouping_and_flush(monkeypatch):
    """ObservationQueue host logic (no kernel runs here: the facade is replaced by a recorder): jobs are
    grouped by statistic kind / bin count / symmetry, flushed as ONE call per group, automatically once the
    pending bytes exceed the bound, and `recorder` keeps (observer, tensor) pairs for reuse_activations."""
    from ppq_amd import observer as obs
    calls = []

    class FakeCUDA:
        @staticmethod
        def MinMax_T_Slots_Multi(values, slots): calls.append(('minmax', len(values)))
        @staticmethod
        def Histogram_T_Rows_Multi(values, rows, scales): calls.append(('hist_sym', len(values), rows[0].shape[1], list(scales)))
        @staticmethod
        def Histogram_Asymmetric_T_Rows_Multi(mins, maxs, values, rows): calls.append(('hist_asym', len(values), list(mins), list(maxs)))
        @staticmethod
        def Quantile_Multi(values, q, dests, hints=None): calls.append(('quantile', len(values), q))
    monkeypatch.setattr(obs, 'CUDA', FakeCUDA)
    q = obs.ObservationQueue(max_pending_bytes=4 * 1000 * 3 + 1)          # room for three 1000-element tensors
    t = [torch.zeros(1000) for _ in range(8)]
    rows2048, rows512 = torch.zeros(4, 2048, dtype=torch.int32), torch.zeros(4, 512, dtype=torch.int32)
    q.add_minmax(t[0], torch.zeros(4, 2)); q.add_hist(t[1], rows2048, False, 0.5); q.add_hist(t[2], rows512, False, 0.25)
    assert len(q) == 3 and not calls
    q.add_hist(t[3], rows2048, True, -1.0, 2.0)                           # 4th tensor: over the bound -> flush
    assert len(q) == 0 and q.launches == 4
    assert sorted(c[0] for c in calls) == ['hist_asym', 'hist_sym', 'hist_sym', 'minmax']
    assert ('hist_sym', 1, 2048, [0.5]) in calls and ('hist_sym', 1, 512, [0.25]) in calls
    assert ('hist_asym', 1, [-1.0], [2.0]) in calls
    calls.clear()
    d1, d2 = q.add_quantile(t[4], 0.9999), q.add_quantile(t[5], 0.9999)
    q.add_quantile(t[6], 0.99)
    assert d1.shape == (2,) and d1.data_ptr() != d2.data_ptr() and len(q) == 3
    q.flush()
    assert sorted(calls) == [('quantile', 1, 0.99), ('quantile', 2, 0.9999)] and len(q) == 0
    q.flush()                                                             # empty: no calls
    assert len(calls) == 2


def test_dense_layout_rule():
    """ffi._dense: which tensors the kernels may stream in storage order without a layout copy."""
    from ppq_amd.ffi import _dense
    x = torch.randn(2, 6, 5, 4)
    xcl = x.contiguous(memory_format=torch.channels_last)
    assert _dense(x) is x and _dense(xcl) is xcl                          # per-tensor: any dense layout
    assert _dense(xcl, 0) is xcl and _dense(xcl, -4) is xcl               # per-channel on the outermost axis
    y = _dense(xcl, 1)                                                    # per-channel on axis 1: NCHW copy
    assert y is not xcl and y.is_contiguous() and torch.equal(y, x)
    v = x[:, ::2]                                                         # strided view: copy
    assert _dense(v).is_contiguous() and torch.equal(_dense(v), v)
    x5 = torch.randn(2, 3, 4, 5, 6).contiguous(memory_format=torch.channels_last_3d)
    assert _dense(x5) is x5


def test_harness_topologies():
    """The synthetic topologies the measurements run on: ResNet-50 (53 Conv + 1 Gemm, 25.5 M parameters
    with BN folded) and ViT-B/16 (86.6 M parameters, 197 tokens); both run an FP32 forward on the CPU."""
    from ppq_amd import harness
    g = harness.resnet50_graph(seed=0)
    types = [op.type for op in g.operations.values()]
    assert types.count('Conv') == 53 and types.count('Gemm') == 1 and len(types) == 122
    n_params = sum(v.value.numel() for v in g.variables.values() if v.is_parameter)
    assert 25.4e6 < n_params < 25.7e6
    y = harness.TorchExecutor(g, 'cpu').forward(torch.rand(1, 3, 64, 64))[0]
    assert y.shape == (1, 1000) and torch.isfinite(y).all()
    harness.quantize_graph(g, 'kl', hist_bins=2048)
    observed = sum(1 for op in g.operations.values() for c, v in op.config_with_variable
                   if not v.is_parameter and c.state.value == 1)
    assert observed == 72                                                  # the 72 activation tensors of the bench
    v = harness.vit_graph(seed=0, depth=2, dim=64, heads=4, mlp_dim=128, patch=16, image=64, num_classes=10)
    y = harness.TorchExecutor(v, 'cpu').forward(torch.randn(2, 3, 64, 64))[0]
    assert y.shape == (2, 10) and torch.isfinite(y).all()
    full = harness.vit_graph(seed=0)
    assert abs(sum(p.value.numel() for p in full.variables.values() if p.is_parameter) - 86_567_656) == 0
    harness.quantize_graph_fp8(full)
    acts = sum(1 for op in full.operations.values() for c, p in op.config_with_variable if not p.is_parameter and c.state.value == 1)
    weights = sum(1 for op in full.operations.values() for c, p in op.config_with_variable if p.is_parameter and c.state.value == 1)
    assert (acts, weights) == (98, 50)


def test_bench_spawns_its_own_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2 ...` with no launcher around it: bench.py re-executes itself under
    torch.distributed.run on 127.0.0.1 and rank 0 prints ONE JSON line with n_gpus == 2.  The CPU-only
    `--host-selftest` variant exercises spawn + rendezvous + the flat per-phase merge over gloo (the data
    path itself needs a GPU: tests/test_gpu_calibration.py::test_bench_two_ranks_on_one_gpu)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--host-selftest'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['merged_counts_ok'] is True
    assert [m['collectives'] for m in out['merge']] == [1, 1]            # ONE all-reduce per phase
    assert out['merge'][1]['sum_int32_bytes'] == 72 * 2048 * 4
    # a world size that contradicts --gpus is refused, not silently run as 1 rank
    env2 = dict(env, WORLD_SIZE='1', RANK='0')
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--host-selftest'],
                        capture_output=True, text=True, timeout=120, env=env2)
    assert r2.returncode != 0 and 'WORLD_SIZE=1' in (r2.stderr + r2.stdout)


def test_merge_refuses_mismatched_layouts():
    """An observer that saw no batch on one rank declares nothing: the merge must fail loudly on every
    rank instead of all-reducing buffers of different length."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_mismatch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs: p.join(timeout=60)
    assert all('different statistics layouts' in msg for _, msg in res), res


def _mismatch_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ppq_amd.distributed import merge_observers

    class FakeObserver:
        def __init__(self, bufs): self.bufs = bufs
        def reducible(self): return self.bufs

    class FakeRowSet:                       # the isotone observer's protocol: rows are gathered, not reduced
        def __init__(self, rows): self.rows, self.got = rows, None
        def reducible(self): return []
        def gatherable(self): return [self.rows]
        def take_gathered(self, merged): self.got = merged[0]
    obs = [FakeObserver([(torch.zeros(64, dtype=torch.int32), 'sum')])]
    if rank == 0: obs.append(FakeObserver([(torch.zeros(64, dtype=torch.int32), 'sum')]))    # rank 1 saw no batch for it
    try:
        merge_observers(obs); msg = 'no error'
    except RuntimeError as e:
        msg = str(e)
    q.put((rank, msg))
    dist.destroy_process_group()


def test_block_builder_blocks_satisfy_the_reference_definition():
    """ppq_amd/blocks.py against the DEFINITION of a TrainableBlock (training.py:229-242), checked by brute
    force on the YOLOv6-s-like and ResNet-50 topologies: M holds exactly the operations on paths S -> E, no
    value enters M except through S, none leaves except through E, depth(E) - depth(S) <= limit; computing
    ops are covered exactly once; known shapes: a Conv-Relu chain, the SPPF fan-out closing at its Concat,
    a residual branch that cannot extend past the Add."""
    from ppq_amd import harness
    from ppq_amd.blocks import BlockBuilder, downstream_operations, split_graph_into_blocks, upstream_operations
    for build, limit in ((harness.yolov6s_graph, 5), (harness.yolov6s_graph, 1), (harness.resnet50_graph, 4), (harness.small_cnn_graph, 3)):
        g = build(seed=0)
        harness.quantize_graph(g, 'minmax')
        blocks = split_graph_into_blocks(g, None, limit)
        builder = BlockBuilder(g)
        covered = [o.name for b in blocks for o in b.rps]
        assert len(covered) == len(set(covered))
        computing = [o.name for o in g.operations.values() if o.type in ('Conv', 'Gemm')]
        assert all(c in covered for c in computing)
        for b in blocks:
            names = {o.name for o in b.rps}
            assert b.rps[0] is b.sp and b.rps[-1] is b.ep
            assert builder.depth[b.ep.name] - builder.depth[b.sp.name] <= limit
            # forward reachability from S restricted to ops that reach E == M
            def reach(start, step):
                seen, todo = {start.name}, [start]
                while todo:
                    for nxt in step(todo.pop()):
                        if nxt.name not in seen: seen.add(nxt.name); todo.append(nxt)
                return seen
            on_paths = reach(b.sp, downstream_operations) & reach(b.ep, upstream_operations)
            assert on_paths == names, (str(b), sorted(on_paths ^ names))
            for o in b.rps:
                if o is not b.sp:
                    assert all(u.name in names for u in upstream_operations(o))
                    assert not any((not v.is_parameter) and v.source_op is None for v in o.inputs)
                if o is not b.ep:
                    assert all(d.name in names for d in downstream_operations(o))
                    assert not any(v.name in g.outputs for v in o.outputs)
    g = harness.yolov6s_graph(seed=0); harness.quantize_graph(g, 'minmax')
    by_sp = {b.sp.name: b for b in split_graph_into_blocks(g, None, 5)}
    assert [o.type for o in by_sp['stem1'].rps] == ['Conv', 'Relu'] * 3
    assert by_sp['sppf_in20'].ep.name == 'sppf_cat' and len(by_sp['sppf_in20'].rps) == 6
    g = harness.resnet50_graph(seed=0); harness.quantize_graph(g, 'minmax')
    by_sp = {b.sp.name: b for b in split_graph_into_blocks(g, None, 4)}
    assert len(by_sp['down5'].rps) == 1 and by_sp['conv2'].ep.name == 'conv4'


def test_isotone_observer_equals_reference_goldens(golden_dir):
    """OBSERVER_TABLE['isotone'] (observer/order.py) against 12 results rendered by the reference's TorchIsotoneObserver
    (tests/golden/make_golden.py::gen_isotone): multi-batch, single row, 3-D, class axis in the middle, logits with
    negative entries, the no-candidate fall-back to min-max; symmetric and asymmetric -- scale and offset bit for bit.
    The observer launches no kernel (top-2 via torch, the interval sweep on the host), so this runs on the CPU."""
    import torch
    from ppq_amd.core import OBSERVER_ISOTONE_OBSERVER_AXIS, LinearQuantizationConfig
    from ppq_amd.observer import OBSERVER_TABLE
    z = np.load(os.path.join(golden_dir, 'isotone.npz'))
    var = type('V', (), {'name': 'x', 'is_parameter': False})()
    for k in range(int(z['iso_n'])):
        sym, axis, qmin, qmax = (int(v) for v in z[f'iso_{k}_meta'])
        cfg = LinearQuantizationConfig(symmetrical=bool(sym), quant_min=qmin, quant_max=qmax, num_of_bits=8, calibration='isotone')
        cfg.detail[OBSERVER_ISOTONE_OBSERVER_AXIS] = axis
        ob = OBSERVER_TABLE['isotone'](var, cfg)
        for i in range(int(z[f'iso_{k}_n'])): ob.observe(torch.from_numpy(z[f'iso_{k}_x{i}']))
        ob.render_quantization_config()
        assert int(getattr(cfg.state, 'value', cfg.state)) == 4
        assert np.array_equal(cfg.scale.reshape(-1).numpy().view(np.uint32), z[f'iso_{k}_scale'].view(np.uint32)), (k, cfg.scale, z[f'iso_{k}_scale'])
        assert np.array_equal(cfg.offset.reshape(-1).numpy(), z[f'iso_{k}_offset']), (k, cfg.offset, z[f'iso_{k}_offset'])


def test_block_split_equals_the_reference_on_resnet50_and_yolov6s():
    """ppq_amd.blocks (one forward sweep) vs the reference's BlockBuilder / split_graph_into_blocks
    (algorithm/training.py:191-315, optim/training.py:177-222) on the ResNet-50 topology -- residual fan-out / fan-in,
    down-sample branches -- for depth limits 1, 2, 3, 4, 5 and 8: the same blocks in the same order, each with the same
    start, end and member set (the order of members inside a block follows each graph's own topological sort)."""
    import torch
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    from ppq_amd import harness
    from ppq_amd.blocks import split_graph_into_blocks
    RI.load()
    from ppq.quantization.optim.training import TrainingBasedPass
    rg = RI.quantize_reference_topology(RI.to_reference_graph(harness.resnet50_graph(seed=0)))
    hg = harness.resnet50_graph(seed=0)
    harness.quantize_graph(hg, 'minmax')
    p = TrainingBasedPass()
    for limit in (1, 2, 3, 4, 5, 8):
        ref = [(b.sp.name, b.ep.name, frozenset(o.name for o in b.rps)) for b in p.split_graph_into_blocks(rg, rg.topological_sort(), limit)]
        ours = [(b.sp.name, b.ep.name, frozenset(o.name for o in b.rps)) for b in split_graph_into_blocks(hg, hg.topological_sort(), limit)]
        assert ref == ours, (limit, [a for a, b in zip(ref, ours) if a != b][:1], [b for a, b in zip(ref, ours) if a != b][:1])
    assert len(ours) == 20
    # BASELINE config 5's topology (EfficientRep + SPPF fan-out closing at its Concat, Rep-PAN neck with Resize, six heads);
    # graph-only on the reference side (no tracing: Resize is not executable through the conversion)
    rg = RI.quantize_reference_topology(RI.to_reference_graph(harness.yolov6s_graph(seed=0)))
    hg = harness.yolov6s_graph(seed=0)
    harness.quantize_graph(hg, 'minmax')
    order = rg.topological_sort()
    for limit, count in ((1, 56), (2, 32), (4, 27), (5, 27), (8, 20)):
        ref = [(b.sp.name, b.ep.name, frozenset(o.name for o in b.rps)) for b in p.split_graph_into_blocks(rg, order, limit)]
        ours = [(b.sp.name, b.ep.name, frozenset(o.name for o in b.rps)) for b in split_graph_into_blocks(hg, hg.topological_sort(), limit)]
        assert ref == ours and len(ours) == count, (limit, len(ref), len(ours))


def _isotone_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ppq_amd.core import LinearQuantizationConfig
    from ppq_amd.distributed import merge_observers
    from ppq_amd.observer import OBSERVER_TABLE
    g = torch.Generator().manual_seed(55)
    batches = [torch.softmax(torch.randn(16 + 3 * i, 10, generator=g) * 3, dim=-1) for i in range(5)]     # ragged row counts
    cfg = LinearQuantizationConfig(symmetrical=True, quant_min=-128, quant_max=127, num_of_bits=8, calibration='isotone')
    ob = OBSERVER_TABLE['isotone'](type('V', (), {'name': 'x', 'is_parameter': False})(), cfg)
    for i, b in enumerate(batches):
        if i % world == rank: ob.observe(b)
    issued = merge_observers([ob])
    ob.render_quantization_config()
    q.put((rank, issued, float(cfg.scale), float(cfg.offset)))
    dist.destroy_process_group()


def test_isotone_observer_data_parallel_gather_equals_union():
    """Two gloo ranks observe disjoint (ragged) shards with the real isotone observer; after merge_observers (all-gather of
    the top-2 pairs) both render the scale a single process renders from all batches."""
    import torch.multiprocessing as mp
    from ppq_amd.core import LinearQuantizationConfig
    from ppq_amd.observer import OBSERVER_TABLE
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 77) % 2000)
    procs = [ctx.Process(target=_isotone_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs: p.join(timeout=60)
    g = torch.Generator().manual_seed(55)
    batches = [torch.softmax(torch.randn(16 + 3 * i, 10, generator=g) * 3, dim=-1) for i in range(5)]
    cfg = LinearQuantizationConfig(symmetrical=True, quant_min=-128, quant_max=127, num_of_bits=8, calibration='isotone')
    ob = OBSERVER_TABLE['isotone'](type('V', (), {'name': 'x', 'is_parameter': False})(), cfg)
    for b in batches: ob.observe(b)
    ob.render_quantization_config()
    for rank, issued, scale, offset in res:
        assert issued == 2 and scale == float(cfg.scale) and offset == float(cfg.offset), (rank, issued, scale, float(cfg.scale))


@pytest.mark.parametrize('topology,configs,n_weights', [('resnet50_graph', 368, 54), ('small_cnn_graph', 23, 3)])
def test_harness_quantizer_assigns_the_reference_config_states_on_resnet50(topology, configs, n_weights):
    """harness.quantize_graph (TensorRT-style policy + the state edits of QuantizeSimplifyPass / QuantizeFusionPass) vs the
    reference's own dispatcher + TensorRT quantizer + those passes on the ResNet-50 topology (and the small CNN of the tests):
    the same 368 (operation, variable) configs with the same state, policy bits, range and channel axis -- except that the
    reference driver has already run its ParameterQuantizePass (54 weight configs ACTIVATED there, still INITIAL here)."""
    import torch
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    from ppq_amd import harness
    RI.load()
    build = getattr(harness, topology)
    rg, _ = RI.quantize_reference_graph(RI.to_reference_graph(build(seed=0)), 'cpu', torch.rand(1, 3, 64, 64), method='kl')
    hg = build(seed=0)
    harness.quantize_graph(hg, 'kl', hist_bins=2048)

    def table(graph):
        out = {}
        for op in graph.operations.values():
            if not hasattr(op, 'config'): continue
            for i, (c, v) in enumerate(op.config_with_variable):
                out[(op.name, v.name, i)] = (c.state.name, int(getattr(c.policy, '_policy', 0)), c.quant_min, c.quant_max, c.num_of_bits,
                                             c.channel_axis if v.is_parameter and i == 1 else None, v.is_parameter)
        return out
    ours, ref = table(hg), table(rg)
    assert set(ours) == set(ref) and len(ours) == configs
    weights = 0
    for k, o in ours.items():
        r = ref[k]
        if o[-1] and o[0] == 'INITIAL':                       # a weight: rendered there already
            assert r[0] == 'ACTIVATED', (k, o, r); weights += 1
            assert o[1:] == r[1:], (k, o, r)
        else:
            assert o[0] == r[0], (k, o, r)
            if o[0] != 'FP32': assert o[1:] == r[1:], (k, o, r)
    assert weights == n_weights


def test_block_builder_on_random_dags_vs_the_reference_walk():
    """Randomised DAGs (Conv / Relu / Add / Concat, fan-out, dead-end branches, several graph outputs): every block of
    ppq_amd.blocks satisfies the block definition (training.py:229-242), and it equals the reference BlockBuilder's
    block WHENEVER the reference's own block satisfies that definition.  (The reference walk only inspects the final
    end point's producers, so on such graphs it also returns regions with an inner operation fed from outside or
    consumed outside -- 145 of 2812 builds over 150 seeds; this package never does.)"""
    import random
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    from ppq_amd import harness
    from ppq_amd.blocks import BlockBuilder, downstream_operations, upstream_operations
    RI.load()
    from ppq.quantization.algorithm.training import BlockBuilder as ReferenceBuilder

    def random_graph(seed, n_ops=14):
        rnd = random.Random(seed)
        g = harness.BaseGraph(f'dag{seed}')
        x = g.create_variable('input'); g.inputs['input'] = x
        vals = [x]
        for i in range(n_ops):
            kind = rnd.choice(['Conv', 'Conv', 'Relu', 'Add', 'Add', 'Concat'])
            a, b = rnd.choice(vals[-5:]), rnd.choice(vals[-5:])
            if kind == 'Conv':
                w = g.create_variable(f'w{i}', torch.zeros(4, 4, 1, 1), True)
                y = g.create_operation('Conv', f'op{i}', [rnd.choice(vals[-4:]), w, g.create_variable(f'b{i}', torch.zeros(4), True)],
                                       {'strides': 1, 'pads': 0})
            elif kind == 'Relu' or a is b: y = g.create_operation('Relu', f'op{i}', [a])
            else: y = g.create_operation(kind, f'op{i}', [a, b], {'axis': 1} if kind == 'Concat' else {})
            vals.append(y)
        for v in vals[1:]:
            if len(v.dest_ops) == 0: g.outputs[v.name] = v
        return g

    def satisfies_definition(g, sp, ep, names):
        for n in names:
            o = g.operations[n]
            if n != sp and (any(u.name not in names for u in upstream_operations(o))
                            or any((not v.is_parameter) and v.source_op is None for v in o.inputs)): return False
            if n != ep and (any(d.name not in names for d in downstream_operations(o))
                            or any(v.name in g.outputs for v in o.outputs)): return False
        return True

    builds = differs = 0
    for seed in range(40):
        hg = random_graph(seed)
        rg = RI.quantize_reference_topology(RI.to_reference_graph(random_graph(seed)))
        harness.quantize_graph(hg, 'minmax')
        ours, ref = BlockBuilder(hg, hg.topological_sort()), ReferenceBuilder(rg, rg.topological_sort())
        for limit in (1, 2, 4, 8):
            for name, op in hg.operations.items():
                if op.type != 'Conv': continue
                b, r = ours.build(op, limit), ref.build(rg.operations[name], limit)
                mine = (b.sp.name, b.ep.name, frozenset(o.name for o in b.rps))
                theirs = (r.sp.name, r.ep.name, frozenset(o.name for o in r.rps))
                builds += 1
                assert satisfies_definition(hg, *mine), (seed, limit, mine)
                assert ours.depth[mine[1]] - ours.depth[mine[0]] <= limit
                if mine != theirs:
                    differs += 1
                    assert not satisfies_definition(hg, *theirs), (seed, limit, mine, theirs)
    assert builds > 500 and 0 < differs < builds // 5


def test_drop_in_quantile_hint_belongs_to_the_declared_owner():
    """CUDA.Quantile(tensor, q) is stateless in the reference.  A caller that declares itself the owner of the calls it makes
    (``ffi.quantile_hint_owner``; install_into_ppq() wraps the reference's percentile observer in it) gets one threshold hint per
    (owner, device, numel, q), weakly held: two observers of equal shape never share thresholds, nothing declared means no hint
    (every call samples) whatever the call stack looks like, owners nest, and a collected observer releases its hints."""
    import functools
    import gc
    from ppq_amd import ffi
    dev = torch.device('cpu')                      # the bookkeeping is device agnostic; kernels are not involved

    def facade(numel, q): return ffi._owner_quantile_hint(dev, numel, q)      # what Quantile_T(hint='auto') evaluates

    class Observer:
        def observe(self, numel, q=0.9999):
            with ffi.quantile_hint_owner(self):
                return functools.partial(lambda: facade(numel, q))()           # wrappers in between do not matter

        def undeclared(self, numel, q=0.9999): return facade(numel, q)
    a, b = Observer(), Observer()
    ha, hb = a.observe(1000), b.observe(1000)
    assert ha is not None and hb is not None and ha is not hb
    assert a.observe(1000) is ha and a.observe(1001) is not ha and a.observe(1000, 0.999) is not ha
    assert ha.dtype == torch.int32 and ha.numel() == 8 and int(ha.abs().sum()) == 0
    assert facade(1000, 0.9999) is None and a.undeclared(1000) is None        # a `self` on the stack is not a declaration
    with ffi.quantile_hint_owner(a):
        with ffi.quantile_hint_owner(b): assert facade(1000, 0.9999) is hb    # innermost owner
        assert facade(1000, 0.9999) is ha                                      # restored on exit
    assert facade(1000, 0.9999) is None

    def short_lived():
        o = Observer(); o.observe(5)
        return len(ffi._owner_hints)
    inside = short_lived()
    gc.collect()
    assert len(ffi._owner_hints) == inside - 1     # the collected observer took its hints with it


def test_quantile_workspace_sizing_covers_the_layout():
    """ppqhip_quantile_multi_workspace_bytes (a host function) must cover what a launch sequence lays out: the prefix (header,
    prefix arrays, device job table), one record per job and two filter lists per job of clamp(n / 128, 16384, 2^20) keys
    rounded up to 32 -- for any mix of sizes, including > 1024 jobs (several sequences share the prefix)."""
    import ctypes
    from ppq_amd import _lib
    lay = (ctypes.c_int64 * 8)()
    _lib.lib.ppqhip_quantile_debug_layout(lay)
    prefix, words = int(lay[0]), int(lay[1])

    def cap(n): return (min(max(n // 128, 16384), 1 << 20) + 31) // 32 * 32
    rng = np.random.default_rng(0)
    for jobs in ([1], [4096], [1605632], [51380224], [2 ** 31 - 1], list(rng.integers(1, 3_000_000, 72)), [300] * 1100 + [10 ** 6],
                 list(rng.integers(1, 200_000_000, 5))):
        need = prefix + sum(words + 2 * cap(int(n)) for n in jobs) * 4
        got = int(_lib.lib.ppqhip_quantile_multi_workspace_bytes(len(jobs), int(sum(int(n) for n in jobs))))
        assert got >= need, (len(jobs), got, need)
        assert got <= 2 * need + (1 << 20)                                                     # and is not wildly larger
    for n in (1, 1605632, 2 ** 31 - 1):
        assert int(_lib.lib.ppqhip_quantile_workspace_bytes(n)) >= prefix + (words + 2 * cap(n)) * 4 >= 65536   # (isotone's partials fit too)


def test_quantile_hot_path_layout_fits_the_workspace():
    """The two-launch path of one hinted tensor (quantile.hip) lays the SAME workspace out its own way: header + exact histograms
    (zeroed by the filter), one 64-B record, one 256-B head pair and two slots per filter workgroup.  ppqhip_quantile_workspace_bytes
    must cover it for every n, regions must not overlap and must keep the alignment the 16-B loads of the select rely on."""
    import ctypes
    from ppq_amd import _lib
    lay = (ctypes.c_int64 * 8)()
    _lib.lib.ppqhip_quantile_hot_layout(lay)
    words, rec, heads, slots, wgs, stage, zero_end, head_keys = (int(v) for v in lay)
    assert zero_end <= rec and rec + wgs * 16 <= heads and heads + wgs * 2 * head_keys <= slots and slots + wgs * 2 * stage == words
    assert rec % 4 == 0 and heads % 4 == 0 and slots % 4 == 0 and stage % 4 == 0 and head_keys % 4 == 0
    assert zero_end >= 64 + 4096 + 2 * 4096 + 2 * 256                      # header + the three exact histograms
    for n in (1, 262144, 1605632, 51380224, 2 ** 31 - 1):
        assert int(_lib.lib.ppqhip_quantile_workspace_bytes(n)) >= words * 4, n


def test_fast_observers_option_wraps_and_restores_the_reference_method():
    """install_into_ppq(fast_observers=True) replaces TorchMinMaxObserver.observe on the reference's class (no file touched) and
    uninstall_from_ppq() puts the reference's own function back; without a device tensor the wrapper defers to it (the host path
    of the reference is untouched)."""
    if not os.path.isdir('/root/reference'):
        pytest.skip('reference not present on this machine')
    import subprocess
    import sys
    code = r'''
import importlib.machinery, os, sys
from unittest.mock import MagicMock
os.environ['PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION'] = 'python'
for n in ['onnx','onnx.helper','onnx.numpy_helper','onnx.mapping','onnx.onnx_pb','onnx.checker','onnx.external_data_helper','onnx.shape_inference','onnx.version_converter']:
    m = MagicMock(); m.__spec__ = importlib.machinery.ModuleSpec(n, None); m.__path__ = []; sys.modules[n] = m
sys.path.insert(0, '/root/reference'); sys.path.insert(0, %r)
import torch, ppq
import ppq_amd
from ppq.quantization.observer.range import TorchMinMaxObserver, TorchHistObserver
from ppq.core import QuantizationPolicy, QuantizationProperty as QP, QuantizationStates, TensorQuantizationConfig, RoundingPolicy
from ppq.IR import Variable
original = TorchMinMaxObserver.observe
ppq_amd.install_into_ppq()
assert TorchMinMaxObserver.observe is original
ppq_amd.install_into_ppq(fast_observers=True)
assert TorchMinMaxObserver.observe is not original and TorchMinMaxObserver.observe.__wrapped__ is original
assert TorchHistObserver.observe is not TorchMinMaxObserver.observe                     # phase 1 of the hist observer reaches it through super()
cfg = TensorQuantizationConfig(policy=QuantizationPolicy(QP.SYMMETRICAL + QP.LINEAR + QP.PER_TENSOR), rounding=RoundingPolicy.ROUND_HALF_EVEN,
                               num_of_bits=8, quant_min=-128, quant_max=127, observer_algorithm='minmax', state=QuantizationStates.INITIAL)
ob = TorchMinMaxObserver(Variable(name='v'), cfg)
for k in range(3): ob.observe(torch.arange(12, dtype=torch.float32).reshape(3, 4) - 4.0 * k)          # host tensors: the reference's own path
assert len(ob._min_val_collector) == 3 and not hasattr(ob, '_ppq_amd_minmax')
ob.render_quantization_config()
assert abs(float(cfg.scale) - 2 * 11.0 / 255) < 1e-7
ppq_amd.uninstall_from_ppq()
assert TorchMinMaxObserver.observe is original
print('OK')
''' % ROOT
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'OK' in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_to_int_dispatch_and_no_cpu_path():
    """PPQLinearQuant_toInt mirrors the reference's dispatch (qfunction/linear.py:218-238): non-linear configs raise, fewer than
    8 bits raise, and -- like every kernel-backed function here -- a host tensor is never computed on the CPU (with a device it
    is staged through the kernel, tests/test_gpu_kernels.py; without one, as here, it raises); PPQuantFunction_toInt refuses dynamic / non-linear policies (qfunction/__init__.py:47-60)."""
    from ppq_amd import FloatingQuantizationConfig, LinearQuantizationConfig, QuantizationStates, qfunction
    t = torch.zeros(4, 6)
    cfg = LinearQuantizationConfig(symmetrical=True, num_of_bits=8)
    cfg.scale, cfg.offset, cfg.state = torch.ones(1), torch.zeros(1), QuantizationStates.ACTIVATED
    with pytest.raises(RuntimeError, match='no CPU path'): qfunction.PPQLinearQuant_toInt(t, cfg)
    low = LinearQuantizationConfig(symmetrical=True, num_of_bits=4, quant_min=-8, quant_max=7)
    low.scale, low.offset = torch.ones(1), torch.zeros(1)
    with pytest.raises(Exception, match='num of bits is unexpected'): qfunction.PPQLinearQuant_toInt(t, low)
    with pytest.raises(ValueError, match='Non-linear'): qfunction.PPQLinearQuant_toInt(t, FloatingQuantizationConfig())
    with pytest.raises(ValueError): qfunction.PPQuantFunction_toInt(t, FloatingQuantizationConfig())
    with pytest.raises(ValueError): qfunction.PPQuantFunction_toInt(t, LinearQuantizationConfig(dynamic=True))


def test_uninstall_restores_the_reference_observer_table_and_type_test():
    """install_plugins_into_ppq(observers=True) swaps PPQ's OBSERVER_TABLE entries and re-binds the two class names its
    calibration pass type-tests (optim/calibration.py:196); uninstall_from_ppq() must put every one of them back,
    also after a repeated install."""
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    RI.load()
    import ppq.quantization.observer as ref_observer
    import ppq.quantization.optim.calibration as ref_calibration
    from ppq.core import PPQ_CONFIG as REF_CONFIG
    from ppq.core.ffi import CUDA_COMPLIER as REF_COMPLIER

    import ppq_amd
    from ppq_amd import observer as ours
    before_table = dict(ref_observer.OBSERVER_TABLE)
    before = (ref_calibration.TorchHistObserver, ref_calibration.TorchMSEObserver, REF_CONFIG.USING_CUDA_KERNEL)
    try:
        ppq_amd.install_plugins_into_ppq(observers=True)
        ppq_amd.install_plugins_into_ppq(observers=True)                   # twice: the saved state must stay the ORIGINAL one
        assert ref_observer.OBSERVER_TABLE['kl'] is ours.TorchHistObserver and 'kl_channel' in ref_observer.OBSERVER_TABLE
        assert ref_calibration.TorchHistObserver is ours.TorchHistObserver and REF_CONFIG.USING_CUDA_KERNEL is True
        assert REF_COMPLIER.__CUDA_EXTENTION__ is ppq_amd.HIP_EXTENSION
    finally:
        ppq_amd.uninstall_from_ppq()
    assert dict(ref_observer.OBSERVER_TABLE) == before_table and 'kl_channel' not in ref_observer.OBSERVER_TABLE
    assert (ref_calibration.TorchHistObserver, ref_calibration.TorchMSEObserver) == before[:2]
    assert REF_CONFIG.USING_CUDA_KERNEL is False and REF_COMPLIER.__CUDA_EXTENTION__ is None
    ppq_amd.uninstall_from_ppq()                                           # idempotent
    assert dict(ref_observer.OBSERVER_TABLE) == before_table


def test_passive_parameter_pass_equals_the_reference_pass():
    """PassiveParameterQuantizePass (optim/parameters.py:13-153) on the REFERENCE's own quantised ResNet-50-topology graph, the
    bias configs switched to the integer platforms' PASSIVE_INIT policy (PPLQuantizer.py:54-66) in two copies: the reference's
    pass on one, this package's (duck-typed on the reference's IR) on the other -- every bias config ends PASSIVE with the same
    scale (= weight scale x input scale), zero offset; an unquantised input raises PermissionError in both; nothing stays
    PASSIVE_INIT.  Host logic only: runs on the CPU."""
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    from ppq_amd import harness
    from ppq_amd.parameters import PassiveParameterQuantizePass as Ours
    RI.load()
    from ppq.core import QuantizationStates as RS
    from ppq.quantization.optim import PassiveParameterQuantizePass as Ref
    from ppq.quantization.optim import RuntimeCalibrationPass as RefCalibration

    def prepared():
        rg, rex = RI.quantize_reference_graph(RI.to_reference_graph(harness.small_cnn_graph(seed=0)), 'cpu', torch.rand(2, 3, 32, 32), method='minmax')
        gen = torch.Generator().manual_seed(0)
        RefCalibration(method='minmax').optimize(graph=rg, dataloader=[torch.rand(2, 3, 32, 32, generator=gen) for _ in range(8)],
                                                 executor=rex, calib_steps=8, collate_fn=None)
        n = 0
        for op in rg.operations.values():
            if hasattr(op, 'config') and op.type in {'Conv', 'Gemm'} and op.num_of_input == 3:
                b = op.config.input_quantization_config[-1]
                b.num_of_bits, b.quant_max, b.quant_min = 32, 2 ** 31 - 1, -(2 ** 31 - 1)
                b.state = RS.PASSIVE_INIT
                n += 1
        assert n >= 3
        return rg
    a, b = prepared(), prepared()
    Ref().optimize(a)
    p = Ours(); p.optimize(b)
    assert p.unresolved == []
    for (na, oa), (nb, ob) in zip(a.operations.items(), b.operations.items()):
        assert na == nb
        if not hasattr(oa, 'config'): continue
        for ca, cb in zip(oa.config.input_quantization_config, ob.config.input_quantization_config):
            assert ca.state == cb.state, (na, ca.state, cb.state)
            if ca.state == RS.PASSIVE:
                assert torch.equal(ca.scale, cb.scale) and torch.equal(ca.offset, cb.offset), na
        if oa.type in {'Conv', 'Gemm'} and oa.num_of_input == 3:
            i_cfg, w_cfg, b_cfg = ob.config.input_quantization_config
            assert b_cfg.state == RS.PASSIVE and torch.equal(b_cfg.scale, w_cfg.scale * i_cfg.scale)
    # the input of the first operation set back to INITIAL: both refuse
    for g, cls in ((prepared(), Ref), (prepared(), Ours)):
        first = next(op for op in g.operations.values() if hasattr(op, 'config') and op.type == 'Conv')
        first.config.input_quantization_config[0].state = RS.INITIAL
        with pytest.raises(PermissionError): cls().optimize(g)


def test_config_dominance_and_master_follow_the_reference_union_find():
    """TensorQuantizationConfig.dominated_by / master_by (core/quant.py:647-713): a dominated config becomes OVERLAPPED and reads
    its root's scale / offset (chains are compressed), a mastered one becomes PASSIVE (PASSIVE_INIT while the master has no
    scale yet); self-domination is refused."""
    from ppq_amd import LinearQuantizationConfig, QuantizationStates as S
    a, b, c = LinearQuantizationConfig(), LinearQuantizationConfig(), LinearQuantizationConfig()
    a.scale, a.offset = torch.tensor([0.5]), torch.tensor([0.0])
    assert a.dominated_by is a and b.dominated_by is b
    b.dominated_by = a
    c.dominated_by = b
    assert c.dominated_by is a and b.state == S.OVERLAPPED and c.state == S.OVERLAPPED and c.scale is a.scale
    d, e = LinearQuantizationConfig(), LinearQuantizationConfig()
    d.master_by = a
    assert d.state == S.PASSIVE and d.scale is a.scale and d.master_by is a
    e.master_by = LinearQuantizationConfig()
    assert e.state == S.PASSIVE_INIT
    with pytest.raises(ValueError): a.master_by = a
    with pytest.raises(AssertionError): a.dominated_by = 3           # the reference asserts here (quant.py:679)
    with pytest.raises(ValueError): a.dominated_by = c                # c's root is a: the son can not dominate its father
    with pytest.raises(TypeError): a.master_by = 3


def test_vectorised_channel_render_equals_the_scalar_loop():
    """observer.minmax_to_scale_offset_channels (whole arrays) == the reference's per-channel loop over numpy float32 scalars
    (range.py:122-129 -> minmax_to_scale_offset) element for element: scales compared as the float32 values the render stores AND
    as doubles, offsets as integers -- symmetric / asymmetric, int8 / uint8 / int4, ranges that are all-positive, all-negative,
    zero, denormal, huge, below the scale threshold, with a manual threshold override; POWER_OF_2 takes the scalar path."""
    from ppq_amd import LinearQuantizationConfig
    from ppq_amd.core import OBSERVER_MIN_SCALE_MANUL_OVERRIDE
    from ppq_amd.observer import minmax_to_scale_offset, minmax_to_scale_offset_channels
    rng = np.random.default_rng(0)
    n = 20000
    a = (rng.standard_normal(n) * 10 ** rng.uniform(-9, 6, n)).astype(np.float32)
    b = (rng.standard_normal(n) * 10 ** rng.uniform(-9, 6, n)).astype(np.float32)
    mins, maxs = np.minimum(a, b), np.maximum(a, b)
    special = np.array([0, -0.0, 1e-45, -1e-45, 3e38, -3e38, 1e-9, -1e-9, 127.5, -127.5, 0.5, -0.5, 255, 1, 2, 3], np.float32)
    mins = np.concatenate([mins, special, special, np.zeros_like(special), -np.abs(special)])
    maxs = np.concatenate([maxs, special, np.zeros_like(special), special, np.abs(special)])
    for sym, qmin, qmax, pow2, override in ((True, -128, 127, False, None), (False, 0, 255, False, None), (True, -8, 7, False, None),
                                           (False, -128, 127, False, None), (True, -128, 127, False, 1e-3), (False, 0, 255, False, 1e-3),
                                           (True, -128, 127, True, None)):
        cfg = LinearQuantizationConfig(symmetrical=sym, quant_min=qmin, quant_max=qmax, power_of_2=pow2, channel_axis=0)
        if override is not None: cfg.detail[OBSERVER_MIN_SCALE_MANUL_OVERRIDE] = override
        want = [minmax_to_scale_offset(x, y, cfg) for x, y in zip(mins, maxs)]
        got_s, got_o = minmax_to_scale_offset_channels(mins, maxs, cfg)
        assert len(got_s) == len(want) == len(got_o)
        ws = np.array([w[0] for w in want], np.float64); gs = np.array(got_s, np.float64)
        assert np.array_equal(ws, gs), (sym, qmin, np.nonzero(ws != gs)[0][:5])
        assert np.array_equal(ws.astype(np.float32), gs.astype(np.float32))
        assert [int(w[1]) for w in want] == [int(v) for v in got_o], (sym, qmin)


def test_prefix_cache_keeps_the_frontier_only_and_the_values_of_a_full_forward(monkeypatch):
    """blocks.PrefixCache on the CPU (a stand-in fake-quant function: the cache logic is host code): block after block of the
    YOLOv6-s-like graph, the inputs it hands out equal a full forward's bit for bit after each block's weights were changed,
    every operation runs ONCE per batch overall, and after each ``invalidate`` the cache holds a cut of the graph -- not the
    whole prefix (what ``prune`` is for)."""
    from ppq_amd import blocks, harness
    g = harness.yolov6s_graph(seed=0)
    harness.quantize_graph(g, 'minmax')
    ex = harness.TorchExecutor(g, 'cpu')
    ex._default_quant_fn = lambda t, c: torch.round(t * 64.0) / 64.0          # deterministic, state-free
    runs = {}
    real = harness._forward

    def counted(op, x):
        runs[op.name] = runs.get(op.name, 0) + 1
        return real(op, x)
    gen = torch.Generator().manual_seed(0)
    batches = [torch.rand(1, 3, 64, 64, generator=gen) for _ in range(2)]
    blks = blocks.split_graph_into_blocks(g, g.topological_sort(), 5)
    assert len(blks) > 10
    prefix = blocks.PrefixCache(g, ex, batches)
    n_act = sum(1 for v in g.variables.values() if not v.is_parameter)
    held = []
    for k, b in enumerate(blks):
        monkeypatch.setattr(harness, '_forward', counted)
        got = prefix.inputs_of(b)
        monkeypatch.setattr(harness, '_forward', real)
        names = list(got[0])
        for i, batch in enumerate(batches):
            want = [batch if n in g.inputs else ex.forward(batch, [n])[0] for n in names]
            for n, w in zip(names, want): assert torch.equal(got[i][n], w), (k, n)
        with torch.no_grad():                                                  # "train" the block
            for op in b.rps:
                for v in op.inputs:
                    if v.is_parameter and v.value.is_floating_point(): v.value.mul_(1.01)
        prefix.invalidate(b)
        held.append(max(len(c) for c in prefix.values))
    assert max(runs.values()) == len(batches), 'an operation of the prefix ran more than once per batch'
    assert max(held) <= 12 < n_act // 4, (max(held), n_act)            # skip connections into the neck stay alive, the rest goes
    assert prefix.resident_bytes() > 0


def test_pass_constructors_take_the_reference_parameters_in_its_order():
    """Every mirrored optimisation pass can be constructed with the reference's own positional / keyword arguments: its
    parameter names are a prefix of this package's, in the same order with the same simple defaults (what this package adds
    comes after them or is keyword-only), and ``optimize`` accepts the keywords ppq.lib.Pipeline passes."""
    import inspect
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    RI.load()
    import ppq.quantization.optim as RO
    from ppq_amd import bias_correction, calibration, lsq, parameters
    mirrors = [('RuntimeCalibrationPass', calibration), ('IsotoneCalibrationPass', calibration), ('PPLDSPTIReCalibrationPass', calibration),
               ('LearnedStepSizePass', lsq), ('BiasCorrectionPass', bias_correction), ('PassiveParameterQuantizePass', parameters),
               ('ParameterQuantizePass', parameters), ('ParameterBakingPass', parameters)]
    for name, module in mirrors:
        ref = [p for p in inspect.signature(getattr(RO, name).__init__).parameters.values() if p.name != 'self']
        ours = [p for p in inspect.signature(getattr(module, name).__init__).parameters.values() if p.name != 'self']
        assert [p.name for p in ours[:len(ref)]] == [p.name for p in ref], (name, [p.name for p in ours], [p.name for p in ref])
        for r, o in zip(ref, ours):
            assert o.kind == inspect.Parameter.POSITIONAL_OR_KEYWORD, (name, o.name)
            if isinstance(r.default, (int, float, str, bool)) or r.default is None:
                assert o.default == r.default, (name, r.name, r.default, o.default)
        opt = inspect.signature(getattr(module, name).optimize).parameters
        assert any(p.kind == inspect.Parameter.VAR_KEYWORD for p in opt.values()), name      # Pipeline: optimize(graph=..., **kwargs)
        assert list(opt)[1] == 'graph', name


def test_config_helper_methods_answer_like_the_reference():
    """TensorQuantizationConfig.can_export / is_same_scheme / is_revisable / copy, QuantizationPolicy.to_dict and
    QuantizationStates.can_export (ppq/core/quant.py:298-306, 361-364, 601-644, 714-723, 865-896) against the reference's own
    classes over every (state, visibility, scale set / unset) combination."""
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    RI.load()
    import ppq.core as rc
    from ppq_amd import core as oc

    def make(m, state, vis, with_scale, axis=None, bits=8):
        P = m.QuantizationProperty
        pol = m.QuantizationPolicy(P.LINEAR.value + P.SYMMETRICAL.value + (P.PER_CHANNEL.value if axis is not None else P.PER_TENSOR.value))
        c = m.TensorQuantizationConfig(policy=pol, rounding=m.RoundingPolicy.ROUND_HALF_EVEN, num_of_bits=bits, quant_min=-128,
                                       quant_max=127, scale=torch.ones(2) if with_scale else None,
                                       offset=torch.zeros(2) if with_scale else None, observer_algorithm='minmax',
                                       channel_axis=axis, state=getattr(m.QuantizationStates, state),
                                       visibility=getattr(m.QuantizationVisibility, vis))
        return c
    states = ['INITIAL', 'ACTIVATED', 'BAKED', 'OVERLAPPED', 'PASSIVE_INIT', 'PASSIVE', 'PASSIVE_BAKED', 'FP32', 'SOI', 'DEQUANTIZED', 'DEACTIVED']
    assert {m.name: m.value for m in rc.QuantizationStates} == {m.name: m.value for m in oc.QuantizationStates}
    for state in states:
        assert rc.QuantizationStates.can_export(getattr(rc.QuantizationStates, state)) == \
            oc.QuantizationStates.can_export(getattr(oc.QuantizationStates, state)), state
        for vis in ('FORCE_EXPORT', 'EXPORT_WHEN_ACTIVE', 'INTERNAL'):
            for with_scale in (False, True):
                r, o = make(rc, state, vis, with_scale), make(oc, state, vis, with_scale)
                for overlapped in (False, True):
                    assert r.can_export(overlapped) == o.can_export(overlapped), (state, vis, with_scale, overlapped)
                assert r.is_revisable() == o.is_revisable(), state
                rt, ot = r.copy(), o.copy()
                assert (rt == r) == (ot == o) and not (ot == o)                      # a copy is a NEW config
                assert ot.state == o.state and ot.visibility == o.visibility and ot.detail == o.detail and ot.detail is not o.detail
                if with_scale:                # cloned -- except an OVERLAPPED copy, which reads its dominator (the original), in both
                    assert torch.equal(ot.scale, o.scale) and (ot.scale is o.scale) == (rt.scale is r.scale) == (state == 'OVERLAPPED')
    a = make(oc, 'ACTIVATED', 'EXPORT_WHEN_ACTIVE', True)
    ra = make(rc, 'ACTIVATED', 'EXPORT_WHEN_ACTIVE', True)
    for kw in ({}, {'axis': 1}, {'bits': 4}):
        assert a.is_same_scheme(make(oc, 'INITIAL', 'INTERNAL', False, **kw)) == ra.is_same_scheme(make(rc, 'INITIAL', 'INTERNAL', False, **kw)), kw
    with pytest.raises(TypeError): a.is_same_scheme(3)
    assert a.policy.to_dict() == ra.policy.to_dict()
    # an OVERLAPPED copy keeps reading its dominator's scale, like the reference's
    b = make(oc, 'ACTIVATED', 'EXPORT_WHEN_ACTIVE', True)
    o = make(oc, 'ACTIVATED', 'EXPORT_WHEN_ACTIVE', False)
    o.dominated_by = b
    assert o.copy().scale is b.scale and not o.is_revisable() and b.is_revisable()


def test_random_dominance_sequences_match_the_reference_config_class():
    """The union-find of TensorQuantizationConfig driven by the same random sequence of ``dominated_by =`` / ``master_by =``
    assignments on this package's configs and on the reference's own (core/quant.py:596-749): after every step the roots,
    states and the scale each config resolves to are the same, and the same assignments are refused."""
    import random
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    RI.load()
    import ppq.core as rc
    from ppq_amd import core as oc
    for seed in range(30):
        rnd = random.Random(seed)
        n = rnd.randint(3, 9)

        def make(m):
            P = m.QuantizationProperty
            pol = m.QuantizationPolicy(P.LINEAR.value + P.SYMMETRICAL.value + P.PER_TENSOR.value)
            return [m.TensorQuantizationConfig(policy=pol, rounding=m.RoundingPolicy.ROUND_HALF_EVEN, num_of_bits=8, quant_min=-128,
                                               quant_max=127, scale=torch.tensor([float(k + 1)]) if k % 2 == 0 else None,
                                               offset=torch.tensor([0.0]) if k % 2 == 0 else None, observer_algorithm='minmax')
                    for k in range(n)]
        ref, ours = make(rc), make(oc)
        for step in range(3 * n):
            i, j = rnd.randrange(n), rnd.randrange(n)
            attr = 'dominated_by' if rnd.random() < 0.7 else 'master_by'
            # master_by has no cycle check in the reference (two configs mastering each other recurse for ever, in both
            # implementations alike): keep the sequences to the assignments its passes make -- a master that is not below
            if attr == 'master_by' and (i == j or ours[j].dominated_by is ours[i] or ours[i].dominated_by is not ours[i]): attr = 'dominated_by'
            outcomes = []
            for cfgs in (ref, ours):
                try:
                    setattr(cfgs[i], attr, cfgs[j]); outcomes.append('ok')
                except (ValueError, TypeError, AssertionError) as e:
                    outcomes.append(type(e).__name__)
            assert outcomes[0] == outcomes[1], (seed, step, attr, i, j, outcomes)
            for k in range(n):
                r_root, o_root = ref[k].dominated_by, ours[k].dominated_by
                assert ref.index(r_root) == ours.index(o_root), (seed, step, k)
                assert ref[k].state.name == ours[k].state.name, (seed, step, k, ref[k].state, ours[k].state)
                rs, os_ = ref[k].scale, ours[k].scale
                assert (rs is None) == (os_ is None) and (rs is None or torch.equal(rs, os_)), (seed, step, k)


def test_policy_validity_and_observer_factory_dispatch_match_the_reference():
    """Every combination of quantisation properties is accepted / refused by QuantizationPolicy exactly as the reference's
    class does (core/quant.py:256-287), and for every valid policy x observer algorithm name (incl. capitalised, unknown and
    missing ones) TensorObserverFactroy.build_observer returns an observer of the same class name or raises the same exception
    type (observer/__init__.py:15-38, the constructors' policy checks in range.py / floating.py / order.py)."""
    import itertools
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    RI.load()
    import ppq.core as rc
    import ppq.quantization.observer as ro
    from ppq_amd import core as oc
    from ppq_amd import observer as oo

    class Var:
        name, is_parameter, value = 'v', False, None

    def bits(m, gran, sym, kind, pow2, dyn):
        P = m.QuantizationProperty
        return (getattr(P, kind).value + getattr(P, sym).value + getattr(P, gran).value + (P.POWER_OF_2.value if pow2 else 0)
                + (P.DYNAMIC.value if dyn else 0))
    checked = 0
    for gran, sym, kind, pow2, dyn in itertools.product(['PER_TENSOR', 'PER_CHANNEL'], ['SYMMETRICAL', 'ASYMMETRICAL'],
                                                        ['LINEAR', 'FLOATING'], [False, True], [False, True]):
        verdict = []
        for m in (rc, oc):
            try: m.QuantizationPolicy(bits(m, gran, sym, kind, pow2, dyn)); verdict.append('ok')
            except Exception as e: verdict.append(type(e).__name__)
        assert verdict[0] == verdict[1], (gran, sym, kind, pow2, dyn, verdict)
        if verdict[0] != 'ok': continue
        for alg in ['minmax', 'kl', 'percentile', 'mse', 'isotone', 'constant', 'floating', 'Minmax', 'KL', 'nonsense', None]:
            out = []
            for m, o in ((rc, ro), (oc, oo)):
                cfg = m.TensorQuantizationConfig(policy=m.QuantizationPolicy(bits(m, gran, sym, kind, pow2, dyn)),
                                                 rounding=m.RoundingPolicy.ROUND_HALF_EVEN, num_of_bits=8, quant_min=-128, quant_max=127,
                                                 observer_algorithm=alg, channel_axis=0 if gran == 'PER_CHANNEL' else None,
                                                 exponent_bits=4 if kind == 'FLOATING' else 0)
                try: out.append(type(o.TensorObserverFactroy.build_observer(Var(), cfg)).__name__)
                except Exception as e: out.append('!' + type(e).__name__)
            assert out[0] == out[1], (gran, sym, kind, pow2, dyn, alg, out)
            checked += 1
    assert checked >= 150


def test_lsq_delegator_trainability_rules_match_the_reference():
    """LSQDelegator.__init__ / trainable_tensors / withdraw (algorithm/training.py:318-376) is host logic: for every state x
    policy (symmetric / asymmetric, power-of-2, floating) x {parameter, activation} x the three trainable flags x a dominated
    config, this package's delegator decides what is trainable, what is backed up and what ``withdraw`` restores exactly as
    the reference's does (same flags, same tensors by position, same restored values)."""
    import itertools
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    RI.load()
    import ppq.core as rc
    from ppq.quantization.algorithm.training import LSQDelegator as RefDelegator
    from ppq_amd import core as oc
    from ppq_amd.lsq import LSQDelegator as OurDelegator

    class Var:
        def __init__(self, is_parameter): self.name, self.is_parameter, self.value = 'v', is_parameter, torch.arange(6.).reshape(2, 3)

    def config(m, state, sym, kind, pow2, dominated, with_scale):
        P = m.QuantizationProperty
        p = getattr(P, kind).value + getattr(P, sym).value + P.PER_TENSOR.value + (P.POWER_OF_2.value if pow2 else 0)
        c = m.TensorQuantizationConfig(policy=m.QuantizationPolicy(p), rounding=m.RoundingPolicy.ROUND_HALF_EVEN, num_of_bits=8,
                                       quant_min=-128, quant_max=127, exponent_bits=4 if kind == 'FLOATING' else 0,
                                       scale=torch.tensor([0.5]) if with_scale else None, offset=torch.tensor([3.0]) if with_scale else None,
                                       observer_algorithm='minmax', state=getattr(m.QuantizationStates, state))
        if dominated:
            boss = m.TensorQuantizationConfig(policy=m.QuantizationPolicy(p), rounding=m.RoundingPolicy.ROUND_HALF_EVEN, num_of_bits=8,
                                              quant_min=-128, quant_max=127, exponent_bits=4 if kind == 'FLOATING' else 0,
                                              scale=torch.tensor([0.25]), offset=torch.tensor([1.0]), observer_algorithm='minmax',
                                              state=m.QuantizationStates.ACTIVATED)
            c.dominated_by = boss
        return c
    n = 0
    for state, (sym, kind, pow2), is_param, dominated, with_scale, flags in itertools.product(
            ['INITIAL', 'ACTIVATED', 'BAKED', 'PASSIVE', 'PASSIVE_INIT', 'FP32'],
            [('SYMMETRICAL', 'LINEAR', False), ('ASYMMETRICAL', 'LINEAR', False), ('SYMMETRICAL', 'LINEAR', True),
             ('SYMMETRICAL', 'FLOATING', True)], [False, True], [False, True], [False, True],
            [(True, True, True), (False, True, True), (True, False, True), (True, True, False)]):
        made = []
        for m, D in ((rc, RefDelegator), (oc, OurDelegator)):
            cfg, var = config(m, state, sym, kind, pow2, dominated, with_scale), Var(is_param)
            d = D(cfg, var, is_parameter_trainable=flags[0], is_scale_trainable=flags[1], is_offset_trainable=flags[2])
            made.append((d, cfg, var))
        (r, rcfg, rvar), (o, ocfg, ovar) = made
        key = (state, sym, kind, pow2, is_param, dominated, with_scale, flags)
        assert (r.is_scale_trainable, r.is_offset_trainable, r.passive, r.is_parameter) == \
               (o.is_scale_trainable, o.is_offset_trainable, o.passive, o.is_parameter), key
        for a in ('scale_backup', 'offset_backup', 'param_backup'):
            x, y = getattr(r, a), getattr(o, a)
            assert (x is None) == (y is None) and (x is None or torch.equal(x, y)), (key, a)

        def role(t, cfg, var): return 'offset' if t is cfg.offset else 'scale' if t is cfg.scale else 'value' if t is var.value else '?'
        assert [role(t, rcfg, rvar) for t in r.trainable_tensors()] == [role(t, ocfg, ovar) for t in o.trainable_tensors()], key
        for d, cfg, var in made:                                  # "train", then withdraw
            with torch.no_grad():
                for t in d.trainable_tensors(): t.add_(1.0)
            d.withdraw(); d.finalize()
        for t_r, t_o in ((rcfg.scale, ocfg.scale), (rcfg.offset, ocfg.offset), (rvar.value, ovar.value)):
            assert (t_r is None) == (t_o is None) and (t_r is None or torch.equal(t_r, t_o)), key
        n += 1
    assert n == 6 * 4 * 2 * 2 * 2 * 4


def test_ctypes_prototypes_agree_with_the_header_parameter_by_parameter():
    """include/ppq_hip.h is the contract, ppq_amd/_lib.PROTOTYPES the binding: for every declared entry point the binding has
    the same number of parameters, and each is of the same KIND (pointer / integer / float / double) in the same position, with
    the same kind of return value -- an argument added on one side only would shift everything behind it."""
    import ctypes
    import re
    from ppq_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'ppq_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    decls = re.findall(r'\b(int64_t|int|float|double|void|const\s+char\s*\*)\s+(ppqhip_\w+)\s*\(([^;{}]*?)\)\s*;', text)
    assert len(decls) >= 55

    def kind_of_c(param: str) -> str:
        p = param.strip()
        if '*' in p: return 'ptr'
        base = re.sub(r'\b\w+$', '', p).strip() or p              # drop the parameter name
        base = base.replace('const', '').strip()
        if base in ('float',): return 'f32'
        if base in ('double',): return 'f64'
        if base in ('int', 'int64_t', 'uint32_t', 'int32_t', 'unsigned', 'unsigned int'): return 'int'
        raise AssertionError(f'unrecognised parameter {param!r}')

    def kind_of_ctypes(t) -> str:
        if t is None: return 'void'
        if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, 'contents') or issubclass(t, ctypes._Pointer): return 'ptr'
        if t is ctypes.c_float: return 'f32'
        if t is ctypes.c_double: return 'f64'
        if t in (ctypes.c_int, ctypes.c_int64, ctypes.c_uint32, ctypes.c_int32): return 'int'
        raise AssertionError(f'unrecognised ctypes type {t!r}')
    seen = set()
    for ret, name, params in decls:
        assert name in _lib.PROTOTYPES, f'{name} is declared in the header but has no ctypes prototype'
        res, args = _lib.PROTOTYPES[name]
        plist = [p for p in params.split(',') if p.strip() and p.strip() != 'void']
        assert len(plist) == len(args), (name, len(plist), len(args))
        assert [kind_of_c(p) for p in plist] == [kind_of_ctypes(a) for a in args], name
        want = 'ptr' if '*' in ret else {'int': 'int', 'int64_t': 'int', 'float': 'f32', 'double': 'f64', 'void': 'void'}[ret]
        assert kind_of_ctypes(res) == want, (name, ret, res)
        seen.add(name)
    assert seen == set(_lib.PROTOTYPES), sorted(set(_lib.PROTOTYPES) - seen)


def test_register_calibration_observer_extends_the_factory():
    """ppq/lib/extension.py:76-93 mirrored: a user observer class registered under a name is what the factory builds for
    configs naming that algorithm (case-insensitively); non-classes and non-observers are refused with TypeError."""
    from ppq_amd import LinearQuantizationConfig
    from ppq_amd import observer as O

    class Mine(O.BaseTensorObserver):
        def observe(self, value): pass
        def render_quantization_config(self): pass
    saved = dict(O.OBSERVER_TABLE)
    try:
        O.register_calibration_observer('MyAlgo', Mine)
        cfg = LinearQuantizationConfig(calibration='myalgo')
        assert type(O.TensorObserverFactroy.build_observer('v', cfg)) is Mine
        cfg.observer_algorithm = 'MYALGO'
        assert type(O.TensorObserverFactroy.build_observer('v', cfg)) is Mine
        with pytest.raises(TypeError): O.register_calibration_observer('x', Mine('v', cfg))
        with pytest.raises(TypeError): O.register_calibration_observer('x', dict)
    finally:
        O.OBSERVER_TABLE.clear(); O.OBSERVER_TABLE.update(saved)


def test_lib_pipeline_and_quant_stub_host_logic():
    """ppq_amd.lib (the on-path slice of ppq.lib): Pipeline runs its passes in order with the caller's keywords, ``at_front``
    prepends, non-passes are refused; a quant stub refuses to render before it saw data, ParameterQuant refuses a non-tensor;
    PFL.Observer builds what the config names."""
    import ppq_amd.lib as PFL
    from ppq_amd.calibration import QuantizationOptimizationPass
    from ppq_amd.observer import TorchHistObserver
    log = []

    class Step(QuantizationOptimizationPass):
        def __init__(self, tag): super().__init__(name=f'step {tag}'); self.tag = tag
        def optimize(self, graph, **kwargs): log.append((self.tag, graph, sorted(kwargs)))
    a, b, c = Step('a'), Step('b'), Step('c')
    pipe = PFL.Pipeline([a, b])
    assert pipe.append_optimization_to_pipeline(c, at_front=True) is pipe and len(pipe) == 3 and a in pipe
    pipe.optimize(graph='G', verbose=False, dataloader=[1], executor=None, calib_steps=8)
    assert [t for t, _, _ in log] == ['c', 'a', 'b'] and all(g == 'G' and k == ['calib_steps', 'dataloader', 'executor'] for _, g, k in log)
    assert pipe.report().count('\n') == 3 and 'step c' in pipe.report().splitlines()[0]
    with pytest.raises(AssertionError): PFL.Pipeline([a, 'not a pass'])
    with pytest.raises(AssertionError): 3 in pipe
    cfg = PFL.LinearQuantizationConfig(calibration='kl')
    assert isinstance(PFL.Observer(cfg), TorchHistObserver)
    stub = PFL.TensorQuant(cfg)
    with pytest.raises(PermissionError): stub.render()
    with pytest.raises(TypeError): PFL.ParameterQuant(cfg, [1.0, 2.0])
    seen = []
    stub.delegator = lambda t, c: seen.append(c) or t + 1
    assert float(stub(torch.zeros(1))) == 1.0 and seen == [cfg] and stub.delegator is not None


def test_passive_parameter_pass_on_clip_and_pad_matches_the_reference_on_seven_platforms():
    """PassiveParameterQuantizePass's Clip / Pad / bias branches (optim/parameters.py:13-153) on a Conv -> Clip(min, max) ->
    Pad(pads, value) -> Relu graph built with the reference's graph API and quantised by SEVEN of its quantizers: the reference's
    pass on one copy, this package's (duck-typed) on the other -- the same states, scales and visibilities (members of the
    REFERENCE's enum, so that its exporters' ``can_export`` answers the same), and the same PermissionError when the input of
    the Clip / the Pad / the Conv has not been quantised."""
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    RI.load()
    import ppq.lib as PFL
    from ppq import BaseGraph, QuantizationProperty, TargetPlatform
    from ppq.core import NetworkFramework
    from ppq.core import QuantizationStates as RS
    from ppq.quantization.optim import PassiveParameterQuantizePass as Ref
    from ppq_amd.parameters import PassiveParameterQuantizePass as Ours

    def prepared(platform, unquantised):
        g = BaseGraph(name='t', built_from=NetworkFramework.ONNX)
        gen = torch.Generator().manual_seed(0)

        def v(n, val=None, p=False): return g.create_variable(name=n, value=val, is_parameter=p)
        x, w, b, y = v('x'), v('w', torch.randn(4, 3, 3, 3, generator=gen), True), v('b', torch.randn(1, 4, generator=gen), True), v('y')
        g.create_operation(op_type='Conv', name='conv', attributes={'kernel_shape': [3, 3], 'strides': [1, 1], 'pads': [1, 1, 1, 1],
                                                                    'dilations': [1, 1], 'group': 1}, inputs=[x, w, b], outputs=[y])
        lo, hi, c = v('lo', torch.tensor(0.0), True), v('hi', torch.tensor(6.0), True), v('c')
        g.create_operation(op_type='Clip', name='clip', attributes={}, inputs=[y, lo, hi], outputs=[c])
        pads, val, p = v('pads', torch.tensor([0, 0, 1, 1, 0, 0, 1, 1]), True), v('val', torch.tensor(0.0), True), v('p')
        g.create_operation(op_type='Pad', name='pad', attributes={'mode': 'constant'}, inputs=[c, pads, val], outputs=[p])
        r = v('r')
        g.create_operation(op_type='Relu', name='relu', attributes={}, inputs=[p], outputs=[r])
        g.mark_variable_as_graph_input(x); g.mark_variable_as_graph_output(r)
        quantizer = PFL.Quantizer(platform=platform, graph=g)
        for op in list(g.operations.values()):
            op.platform = platform
            quantizer.quantize_operation(op.name, platform=platform)
        for op in g.operations.values():                            # a stand-in calibration
            for k, (cfg, _) in enumerate(op.config_with_variable):
                if cfg.state == RS.INITIAL and not (unquantised == op.type and k == 0):
                    n = 4 if cfg.policy.has_property(QuantizationProperty.PER_CHANNEL) else 1
                    cfg.scale, cfg.offset, cfg.state = torch.full([n], 0.1 * (k + 1)), torch.zeros(n), RS.ACTIVATED
        return g
    passive = 0
    for name in ('PPL_CUDA_INT8', 'PPL_DSP_INT8', 'SNPE_INT8', 'TRT_INT8', 'OPENVINO_INT8', 'METAX_INT8_C', 'QNN_DSP_INT8'):
        for unquantised in (None, 'Clip', 'Pad', 'Conv'):
            a, b = (prepared(getattr(TargetPlatform, name), unquantised) for _ in range(2))
            outcome = []
            for pass_, g in ((Ref(), a), (Ours(), b)):
                try: pass_.optimize(g); outcome.append(None)
                except PermissionError as e: outcome.append(str(e))
            assert outcome[0] == outcome[1], (name, unquantised, outcome)
            for (na, oa), (_, ob) in zip(a.operations.items(), b.operations.items()):
                for (ca, va), (cb, vb) in zip(oa.config_with_variable, ob.config_with_variable):
                    key = (name, unquantised, na, va.name)
                    assert ca.state is cb.state and ca.visibility is cb.visibility and ca.can_export() == cb.can_export(), key
                    assert (ca.scale is None) == (cb.scale is None) and (ca.scale is None or torch.equal(ca.scale, cb.scale)), key
                    assert va.value is None or va.value.shape == vb.value.shape, key
                    passive += ca.state == RS.PASSIVE
    assert passive >= 40


def test_isotone_pass_marks_the_same_configs_as_the_reference(monkeypatch):
    """IsotoneCalibrationPass's marking step (optim/calibration.py:325-422; the calibration it ends in is stubbed out on both
    sides) on a Gemm -> Softmax -> Softmax graph built with the reference's graph API under four of its quantizers: default
    (every Softmax output that is its own root), named variables with an axis, a missing name, a non-list and a list with a
    non-string -- same states / algorithms / axes on every config afterwards (incl. what was marked BEFORE a bad entry raised),
    same exception type and text."""
    from oracle import reference_import as RI
    if RI.find_reference() is None: pytest.skip('reference not present on this machine')
    RI.load()
    import ppq.lib as PFL
    import ppq.quantization.optim.calibration as rcal
    from ppq import BaseGraph, TargetPlatform
    from ppq.core import NetworkFramework
    from ppq.core import QuantizationStates as RS
    from ppq_amd import calibration as ocal
    monkeypatch.setattr(rcal.RuntimeCalibrationPass, 'optimize', lambda self, graph, **kw: None)
    monkeypatch.setattr(ocal.RuntimeCalibrationPass, 'optimize', lambda self, graph, **kw: None)

    def build(platform):
        g = BaseGraph(name='t', built_from=NetworkFramework.ONNX)

        def v(n, val=None, p=False): return g.create_variable(name=n, value=val, is_parameter=p)
        x, w, b, y, s, s2 = v('x'), v('w', torch.randn(10, 8), True), v('b', torch.randn(10), True), v('y'), v('s'), v('s2')
        g.create_operation(op_type='Gemm', name='fc', attributes={'alpha': 1.0, 'beta': 1.0, 'transA': 0, 'transB': 1}, inputs=[x, w, b], outputs=[y])
        g.create_operation(op_type='Softmax', name='sm', attributes={'axis': 1}, inputs=[y], outputs=[s])
        g.create_operation(op_type='Softmax', name='sm2', attributes={}, inputs=[s], outputs=[s2])
        g.mark_variable_as_graph_input(x); g.mark_variable_as_graph_output(s2)
        quantizer = PFL.Quantizer(platform=platform, graph=g)
        for op in list(g.operations.values()):
            op.platform = platform
            quantizer.quantize_operation(op.name, platform=platform)
        for op in g.operations.values():
            for c, _ in op.config_with_variable:
                if c.state == RS.INITIAL: c.state, c.scale, c.offset = RS.ACTIVATED, torch.tensor([0.1]), torch.tensor([0.0])
        return g

    def table(g):
        return [(op.name, var.name, c.state.name, c.observer_algorithm, c.detail.get('OBSERVER_ISOTONE_OBSERVER_AXIS'))
                for op in g.operations.values() for c, var in op.config_with_variable]
    marked = 0
    for name in ('PPL_CUDA_INT8', 'SNPE_INT8', 'TRT_INT8', 'OPENVINO_INT8'):
        for kw in ({}, {'variables': ['y'], 'axis': 0}, {'variables': ['s', 'y'], 'axis': -1}, {'variables': ['nope']},
                   {'variables': 's'}, {'variables': ['s', 3]}):
            a, b = build(getattr(TargetPlatform, name)), build(getattr(TargetPlatform, name))
            out = []
            for cls, g in ((rcal.IsotoneCalibrationPass, a), (ocal.IsotoneCalibrationPass, b)):
                try: cls(verbose=False, **kw).optimize(g); out.append(None)
                except (TypeError, ValueError) as e: out.append((type(e).__name__, str(e)))
            assert out[0] == out[1] and table(a) == table(b), (name, kw, out)
            marked += sum(1 for row in table(b) if row[3] == 'Isotone')
    assert marked >= 20


def test_calibration_pass_enters_every_executor_through_its_public_forward():
    """RuntimeCalibrationPass._forward_fn: the reference's ``TorchExecutor.forward`` is ``forward_with_gradient`` under
    ``@torch.no_grad()`` (executor/torch.py:365-410; ``@empty_ppq_cache`` decorates only ``tracing_operation_meta``, :579-580), so
    the pass calls ``forward`` on the reference's executor as on any other -- under its own no_grad."""
    import torch
    from ppq_amd.calibration import RuntimeCalibrationPass
    calls = []

    class RefLike:
        _executing_order = []
        def forward(self, inputs, output_names=None, hooks=None): calls.append(('forward', torch.is_grad_enabled()))
        def forward_with_gradient(self, inputs, output_names=None, hooks=None): calls.append(('forward_with_gradient', torch.is_grad_enabled()))
    RefLike.__module__ = 'ppq.executor.torch'

    class Other(RefLike): pass
    Other.__module__ = 'ppq_amd.harness'
    p = RuntimeCalibrationPass(method='minmax')
    with torch.enable_grad():
        for ex in (RefLike(), Other()):
            p._forward(ex, torch.zeros(1), {}, None)
    assert calls == [('forward', False), ('forward', False)]


def test_cuda_mirror_carries_order_preserving_observe_with_the_reference_wiring():
    """ppq/core/ffi.py:257-261: `CUDA.OrderPreservingObserve(tensor)` forwards ONE argument to `RoundingLoss_LC_B` -- every call
    fails in the extension's argument check (TypeError from pybind there, from the Python signature here); nothing in ppq calls it.
    The name exists for surface completeness (missing 4)."""
    import torch
    from ppq_amd import CUDA
    assert callable(CUDA.OrderPreservingObserve)
    with pytest.raises(TypeError):
        CUDA.OrderPreservingObserve(torch.zeros(4))


def test_only_parameter_shaped_tensors_join_the_queued_per_channel_minmax():
    """the multi-tensor per-channel min/max launch takes one wave per ROW, so an activation [N, C, H, W] with short rows
    is served better by the single-tensor kernel (several rows of a channel per wave); only tensors whose channel axis is the
    outermost non-trivial one -- weights [Cout, Cin, kh, kw] on axis 0, a batch-1 activation on axis 1 -- are queued
    (observer.TorchMinMaxObserver.observe asks CUDA.minmax_c_outer_is_one; pure geometry, no kernel)."""
    import torch
    from ppq_amd import CUDA
    cases = [((64, 3, 7, 7), 0, True), ((1000, 2048), 0, True), ((1, 512, 56, 56), 1, True), ((1, 1, 197, 768), 2, True),
             ((8, 512, 56, 56), 1, False), ((2, 197, 768), 2, False), ((32, 27), 1, False), ((5,), 0, True)]
    for shape, axis, want in cases:
        assert CUDA.minmax_c_outer_is_one(torch.zeros(shape), axis) is want, (shape, axis)
    # a parameter-shaped tensor may additionally be `fresh` (overwritten, no seeding) when a channel fits one wave's chunk
    assert CUDA.minmax_c_fresh_ok(torch.zeros(64, 3, 7, 7), 0) and not CUDA.minmax_c_fresh_ok(torch.zeros(8, 512, 56, 56), 1)
    assert not CUDA.minmax_c_fresh_ok(torch.zeros(7, 9001), 0)          # outer == 1 but 9001 > 8192 elements per channel
