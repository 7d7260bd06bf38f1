"""GPU tests of the host layer: observers, batched render, the calibration pass and the fake-quant
functions, compared with the oracle's observer arithmetic replayed on the SAME tensors (exact
float32 scales / integer offsets) and with the scales the reference itself rendered
(tests/golden/observers.npz) where the reference's CPU path shares the rule."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ppq_oracle as O
from oracle.cpu_calibration import ReplayObserver

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def CUDA():
    from ppq_amd import CUDA as C
    return C


def _cfg(alg, sym=True, bits=8, per_channel_axis=None, bins=None, pow2=False, pct=None):
    from ppq_amd import LinearQuantizationConfig
    qmin, qmax = (-(2 ** (bits - 1)), 2 ** (bits - 1) - 1) if sym else (0, 2 ** bits - 1)
    c = LinearQuantizationConfig(symmetrical=sym, quant_min=qmin, quant_max=qmax, num_of_bits=bits, calibration=alg,
                                 channel_axis=per_channel_axis, power_of_2=pow2)
    if bins: c.detail['OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE'] = bins
    if pct: c.detail['OBSERVER_PERCENTILE_MANUL_OVERRIDE'] = pct
    return c


def _batches(relu=False, n=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    bs = [torch.randn(2, 16, 14, 14, generator=g) * (1 + 0.1 * i) + 0.2 for i in range(n)]
    return [torch.relu(b) for b in bs] if relu else bs


def _run_observer(cfg, data, two_phase, batched):
    from ppq_amd.observer import TensorObserverFactroy, render_observers
    ob = TensorObserverFactroy.build_observer('x', cfg)
    for b in data: ob.observe(b.to(DEV))
    render_observers([ob]) if batched else ob.render_quantization_config()
    if two_phase:
        for b in data: ob.observe(b.to(DEV))
        render_observers([ob]) if batched else ob.render_quantization_config()
    return ob


@pytest.mark.parametrize('alg', ['minmax', 'kl', 'mse', 'percentile'])
@pytest.mark.parametrize('sym', [True, False])
@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('batched', [False, True])
def test_observer_matches_oracle_replay(alg, sym, relu, batched):
    if alg == 'kl' and not sym: pytest.skip('KL is symmetric only (range.py:219-220)')
    data = _batches(relu)
    bins = 2048
    cfg = _cfg(alg, sym, bins=bins if alg == 'kl' else None)
    ob = _run_observer(cfg, data, two_phase=alg in ('kl', 'mse'), batched=batched)
    r = ReplayObserver(alg, cfg.quant_min, cfg.quant_max, 8, sym, bins)
    for b in data: r.phase1(b.numpy())
    if alg in ('kl', 'mse'):
        r.end_phase1()
        for b in data: r.phase2(b.numpy())
        assert np.array_equal(ob._hist.cpu().numpy(), r.hist)            # counts exact
    s, o = r.render()
    assert cfg.state.value == 4
    assert np.float32(s) == cfg.scale.cpu().numpy() and np.float32(o) == cfg.offset.cpu().numpy()
    assert cfg.scale.ndim == 0 and cfg.scale.is_cuda                      # same shape convention as range.py:115


def test_minmax_observers_match_reference_golden(golden_dir):
    """End to end against the scales the REFERENCE's TorchMinMaxObserver rendered on these batches."""
    z = np.load(os.path.join(golden_dir, 'observers.npz'))
    from ppq_amd import LinearQuantizationConfig
    for k in range(int(z['minmax_n'])):
        relu, per_channel, sym, qmin, qmax, bits, pow2 = [int(v) for v in z[f'minmax_{k}_meta']]
        data = [torch.from_numpy(np.maximum(b, 0) if relu else b) for b in z['batches']]
        cfg = LinearQuantizationConfig(symmetrical=bool(sym), power_of_2=bool(pow2), quant_min=qmin, quant_max=qmax,
                                       num_of_bits=bits, channel_axis=1 if per_channel else None)
        _run_observer(cfg, data, two_phase=False, batched=bool(k % 2))
        assert np.array_equal(cfg.scale.cpu().numpy(), z[f'minmax_{k}_scale']), k
        assert np.array_equal(cfg.offset.cpu().numpy(), z[f'minmax_{k}_offset']), k


def test_kl_and_mse_scales_vs_reference_golden(golden_dir):
    """From raw data.  The golden scales come from the reference's CPU path, which bins with
    torch.histc (x == max lands in the last bin); the reference's CUDA kernel -- the rule this package
    reproduces -- drops that one element (sort.cu:84-86, clip_outliers defaults to True).  That single
    count can flip the KL arg-min, in the reference itself.  So: the GPU result must ALWAYS equal the
    oracle with the CUDA rule, and must equal the golden CPU-path scale (1e-6 relative) wherever the
    oracle says the two rules agree -- which has to be the large majority of the cases."""
    z = np.load(os.path.join(golden_dir, 'observers.npz'))
    agree = total = 0
    for k in range(int(z['kl_n']) - 1):
        relu, bins, bits, pow2 = [int(v) for v in z[f'kl_{k}_meta']]
        raw = [np.maximum(b, 0) if relu else b for b in z['batches']]
        cfg = _cfg('kl', True, bits, bins=bins, pow2=bool(pow2))
        _run_observer(cfg, [torch.from_numpy(b) for b in raw], two_phase=True, batched=True)
        r = ReplayObserver('kl', cfg.quant_min, cfg.quant_max, bits, True, bins)
        for b in raw: r.phase1(b)
        r.end_phase1()
        for b in raw: r.phase2(b)
        want = O.kl_search(r.hist, r.hist_scale, bits, bool(pow2))[0]
        assert np.float32(want) == np.float32(float(cfg.scale)), k
        total += 1
        if np.float32(want) == pytest.approx(float(z[f'kl_{k}_scale']), rel=1e-6):
            agree += 1
    assert agree >= total - 4, (agree, total)
    agree = total = 0
    for k in range(int(z['mse_n'])):
        relu, sym, bins, qmin, qmax = [int(v) for v in z[f'mse_{k}_meta']]
        if bins != 2048: continue
        raw = [np.maximum(b, 0) if relu else b for b in z['batches']]
        cfg = _cfg('mse', bool(sym))
        _run_observer(cfg, [torch.from_numpy(b) for b in raw], two_phase=True, batched=True)
        r = ReplayObserver('mse', qmin, qmax, 8, bool(sym), 2048)
        for b in raw: r.phase1(b)
        r.end_phase1()
        for b in raw: r.phase2(b)
        s_want, o_want = r.render()
        assert np.float32(s_want) == np.float32(float(cfg.scale)) and float(o_want) == float(cfg.offset), k
        total += 1
        if (float(cfg.scale) == pytest.approx(float(z[f'mse_{k}_scale']), rel=1e-6)
                and float(cfg.offset) == float(z[f'mse_{k}_offset'])):
            agree += 1
    assert agree >= total // 2, (agree, total)     # tiny tensors: one dropped outlier moves the MSE optimum


def test_kl_and_mse_scales_equal_the_reference_on_its_cuda_bin_rule(golden_dir):
    """100 % agreement, no majority vote: tests/golden/observers_cuda_rule.npz holds what the reference's OWN
    hist_to_scale_offset renders from a histogram collected with its CUDA bin rule (and, for MSE, scored by
    its own hist_mse.cc) -- make_golden.py::gen_observers_cuda_rule.  Every GPU-rendered scale / offset must
    equal it (float32 scale to 1e-6 relative as north_star states, integer offset exactly)."""
    z = np.load(os.path.join(golden_dir, 'observers.npz'))
    c = np.load(os.path.join(golden_dir, 'observers_cuda_rule.npz'))
    for k in range(int(c['kl_n'])):
        relu, bins, bits, pow2 = [int(v) for v in c[f'kl_{k}_meta']]
        raw = [np.maximum(b, 0) if relu else b for b in z['batches']]
        cfg = _cfg('kl', True, bits, bins=bins, pow2=bool(pow2))
        _run_observer(cfg, [torch.from_numpy(b) for b in raw], two_phase=True, batched=bool(k % 2))
        assert float(cfg.scale) == pytest.approx(float(c[f'kl_{k}_scale']), rel=1e-6), (k, float(cfg.scale), float(c[f'kl_{k}_scale']))
    for k in range(int(c['mse_n'])):
        relu, sym, bins, qmin, qmax = [int(v) for v in c[f'mse_{k}_meta']]
        raw = [np.maximum(b, 0) if relu else b for b in z['batches']]
        cfg = _cfg('mse', bool(sym))
        if bins != 2048:
            from ppq_amd.observer import TorchMSEObserver, render_observers
            ob = TorchMSEObserver('x', cfg, bins=bins)
            for _ in range(2):
                for b in raw: ob.observe(torch.from_numpy(b).to(DEV))
                render_observers([ob])
        else:
            _run_observer(cfg, [torch.from_numpy(b) for b in raw], two_phase=True, batched=bool(k % 2))
        assert float(cfg.scale) == pytest.approx(float(c[f'mse_{k}_scale']), rel=1e-6), (k, float(cfg.scale), float(c[f'mse_{k}_scale']))
        assert float(cfg.offset) == float(c[f'mse_{k}_offset']), (k, float(cfg.offset), float(c[f'mse_{k}_offset']))


def test_per_channel_minmax_weights_and_qfunction():
    from ppq_amd import qfunction
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    for sym in (True, False):
        cfg = _cfg('minmax', sym, per_channel_axis=0)
        _run_observer(cfg, [w], False, True)
        mins, maxs = O.minmax_c(w.numpy(), 0)
        so = [O.minmax_to_scale_offset(a, b, cfg.quant_min, cfg.quant_max, sym, f32_inputs=True) for a, b in zip(mins, maxs)]
        ws = np.array([v[0] for v in so], np.float32); wo = np.array([v[1] for v in so], np.float32)
        assert np.array_equal(cfg.scale.cpu().numpy(), ws) and np.array_equal(cfg.offset.cpu().numpy(), wo)
        y = qfunction.PPQuantFunction(w.to(DEV), cfg)
        want = O.fq_linear_c(w.numpy(), ws, wo, 0, cfg.quant_min, cfg.quant_max, 0)
        assert np.array_equal(y.cpu().numpy().view(np.uint32), want.view(np.uint32))
        assert y.data_ptr() != w.data_ptr()


def test_qfunction_dispatch_states_and_ste():
    from ppq_amd import FloatingQuantizationConfig, QuantizationStates, qfunction
    x = torch.randn(4, 8, 5, generator=torch.Generator().manual_seed(1)).to(DEV)
    cfg = _cfg('minmax')
    assert qfunction.PPQuantFunction(x, cfg) is x                                    # INITIAL: untouched
    cfg.scale = torch.tensor(0.05, device=DEV); cfg.offset = torch.tensor(0.0, device=DEV)
    cfg.state = QuantizationStates.ACTIVATED
    xr = x.clone().requires_grad_(True)
    y = qfunction.PPQuantFunction(xr, cfg)
    y.sum().backward()
    assert torch.equal(xr.grad, torch.ones_like(x))                                  # straight-through (linear.py:48-50)
    want = O.fq_linear_t(x.cpu().numpy(), [0.05], [0.0], -128, 127, 0)
    assert np.array_equal(y.detach().cpu().numpy().view(np.uint32), want.view(np.uint32))
    cfg.state = QuantizationStates.FP32
    assert qfunction.PPQuantFunction(x, cfg) is x
    f = FloatingQuantizationConfig()
    f.scale = torch.tensor(0.5, device=DEV); f.offset = torch.tensor(0.0, device=DEV); f.state = QuantizationStates.ACTIVATED
    y = qfunction.PPQuantFunction(x, f)
    assert np.array_equal(y.cpu().numpy().view(np.uint32), O.fq_float_t(x.cpu().numpy(), [0.5], [0.0]).view(np.uint32))
    with pytest.raises(PermissionError):
        qfunction.PPQFloatingQuantFunction(x.cpu(), f)
    # dynamic quantisation: min/max of this very tensor
    from ppq_amd import LinearQuantizationConfig
    d = LinearQuantizationConfig(symmetrical=False, dynamic=True, quant_min=0, quant_max=255)
    d.state = QuantizationStates.ACTIVATED
    y = qfunction.PPQuantFunction(x, d)
    mm = O.minmax_t(x.cpu().numpy())
    s, o = O.minmax_to_scale_offset(float(mm[0]), float(mm[1]), 0, 255, False)
    want = O.fq_linear_t(x.cpu().numpy(), [np.float32(s)], [np.float32(o)], 0, 255, 0)
    assert np.array_equal(y.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_floating_observers():
    """DirectMSEObserver ('floating', observer/floating.py:51-143) replayed through the oracle: the device RNG
    is seeded before every observe(), so the very fetches the observer drew (utils/fetch.py:32-50:
    torch.randint on the tensor's device) are re-drawn here, pushed through the oracle's FP8 fake-quant for
    each of the 7 candidate scales, and the observer must have picked the oracle's arg-min (float64 MSE; only
    when the two best candidates are closer than 1e-5 relative -- float32 summation order -- either passes)."""
    from ppq_amd import FloatingQuantizationConfig
    from ppq_amd.observer import DirectMSEObserver, TensorObserverFactroy
    g = torch.Generator().manual_seed(5)
    for case, mult in enumerate((0.01, 1.0, 300.0, 0.2, 30.0)):
        cfg = FloatingQuantizationConfig(calibration='floating')
        ob = TensorObserverFactroy.build_observer('x', cfg)
        assert isinstance(ob, DirectMSEObserver)
        fetched = []
        for k in range(3):
            x = (torch.randn(8, 64, 14, generator=g) * mult).to(DEV)
            torch.manual_seed(1000 * case + k)
            ob.observe(x)
            torch.manual_seed(1000 * case + k)             # re-draw the identical indices
            idx = torch.randint(low=0, high=x.numel(), size=[ob._fetches], device=x.device)
            fetched.append(x.flatten().index_select(0, idx).cpu().numpy())
        ob.render_quantization_config()
        fp = np.concatenate(fetched)
        losses = []
        for scale in DirectMSEObserver.SCALE_CANDIDATES:
            qt = O.fq_float_t(fp, [np.float32(scale)], [np.float32(0)], cfg.exponent_bits, cfg.mantissa_bits,
                              cfg.quant_min, cfg.quant_max, 0)
            losses.append(float(np.mean((qt.astype(np.float64) - fp.astype(np.float64)) ** 2)))
        order = np.argsort(losses, kind='stable')
        best, second = order[0], order[1]
        ok = {DirectMSEObserver.SCALE_CANDIDATES[best]}
        if losses[second] - losses[best] <= 1e-5 * losses[best]: ok.add(DirectMSEObserver.SCALE_CANDIDATES[second])
        assert cfg.state.value == 4 and float(cfg.scale) in ok and float(cfg.offset) == 0.0, (mult, float(cfg.scale), losses)
    cfg = FloatingQuantizationConfig(calibration='constant')
    ob = TensorObserverFactroy.build_observer('x', cfg)
    ob.observe(torch.randn(4, 4).to(DEV)); ob.render_quantization_config()
    assert float(cfg.scale) == 1.0 and float(cfg.offset) == 0.0


@pytest.mark.parametrize('method', ['kl', 'mse', 'minmax', 'percentile'])
def test_runtime_calibration_pass_small_graph(method):
    """The whole pass on a small CNN: every observed tensor's rendered scale equals the oracle's on
    the tensors the observers actually saw; weights are per-channel fake-quantised every forward."""
    from ppq_amd import harness
    from ppq_amd import observer as obs_mod
    from ppq_amd.calibration import RuntimeCalibrationPass
    graph = harness.small_cnn_graph(seed=1)
    harness.quantize_graph(graph, method, hist_bins=2048 if method == 'kl' else None)
    ex = harness.TorchExecutor(graph, DEV)
    harness.ParameterQuantizePass().optimize(graph)
    g = torch.Generator().manual_seed(2)
    batches = [torch.rand(4, 3, 24, 24, generator=g).to(DEV) for _ in range(8)]
    seen = {}
    cls = obs_mod.OBSERVER_TABLE[method]
    orig = cls.observe

    def spy(self, value):
        seen.setdefault(id(self), []).append((getattr(self, '_phase', 'Detecting Minmax'), value.detach().cpu().numpy().copy()))
        return orig(self, value)
    cls.observe = spy
    try:
        p = RuntimeCalibrationPass(method=method)
        obs_first = {}
        real_render = p._render

        def capture_render():
            for op_ob in p._observers.values():
                for ob in op_ob.observers(): obs_first[id(ob)] = ob
            real_render()
        p._render = capture_render
        p.optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
    finally:
        cls.observe = orig
    assert len(obs_first) == 6
    for key, ob in obs_first.items():
        cfg = ob._quant_cfg
        r = ReplayObserver(method, cfg.quant_min, cfg.quant_max, 8, True, 2048)
        for phase, a in seen[key]:
            if phase == 'Detecting Minmax': r.phase1(a)
        if method in ('kl', 'mse'):
            r.end_phase1()
            for phase, a in seen[key]:
                if phase == 'Collating Hist': r.phase2(a)
        s, o = r.render()
        assert cfg.state.value == 4
        if method == 'percentile':   # float32 mean of the per-batch quantiles: reduction order is torch's
            assert float(cfg.scale) == pytest.approx(s, rel=1e-6)
        else:
            assert np.float32(s) == np.float32(float(cfg.scale))
    # a forward through the calibrated graph fake-quantises activations: outputs of an activated
    # per-tensor config lie on its quantisation grid
    y = ex.forward(batches[0], output_names=['c1_relu_out'])[0]
    cfg = graph.operations['c1_relu'].config.output_quantization_config[0]
    codes = (y / cfg.scale).round()
    assert torch.allclose(codes * cfg.scale, y, atol=0, rtol=1e-6) and codes.max() <= 127
    # baking the parameters first is a pure optimisation: identical outputs
    out_before = ex.forward(batches[0])[0].clone()
    from ppq_amd.qfunction import PPQuantFunction
    weights = [(c, v) for op in graph.operations.values() if hasattr(op, 'config') for c, v in op.config_with_variable
               if v.is_parameter and c.state.value in (4, 5)]                      # ACTIVATED / PASSIVE
    expected = {v.name: PPQuantFunction(v.value, c) for c, v in weights}
    baking = harness.ParameterBakingPass()
    baking.optimize(graph)
    assert len(weights) >= 3 and baking.launches == 1 and baking.per_tensor == 0      # all weights: ONE multi-tensor launch
    for c, v in weights:
        assert c.state.value in (2, 7), c.state                                   # BAKED / PASSIVE_BAKED
        assert torch.equal(v.value, expected[v.name]) and v.value.is_contiguous()
    assert torch.equal(ex.forward(batches[0])[0], out_before)


def test_batched_render_single_sync_equals_individual():
    """render_observers over many observers == rendering one by one."""
    from ppq_amd.observer import TensorObserverFactroy, render_observers
    data = _batches(False, n=3, seed=9)
    cfgs_a = [_cfg('kl', bins=2048), _cfg('mse', False), _cfg('minmax', False), _cfg('kl', bits=4, bins=2048), _cfg('mse', True)]
    cfgs_b = [_cfg('kl', bins=2048), _cfg('mse', False), _cfg('minmax', False), _cfg('kl', bits=4, bins=2048), _cfg('mse', True)]
    obs_a = [TensorObserverFactroy.build_observer('x', c) for c in cfgs_a]
    obs_b = [TensorObserverFactroy.build_observer('x', c) for c in cfgs_b]
    for phase in range(2):
        for i, (oa, ob) in enumerate(zip(obs_a, obs_b)):
            for b in data:
                oa.observe((b * (i + 1)).to(DEV)); ob.observe((b * (i + 1)).to(DEV))
        render_observers(obs_a)
        for ob in obs_b: ob.render_quantization_config()
    for ca, cb in zip(cfgs_a, cfgs_b):
        assert ca.state.value == 4 and cb.state.value == 4
        assert torch.equal(ca.scale, cb.scale) and torch.equal(ca.offset, cb.offset)


def test_rows_and_slots_accumulators_match_oracle():
    """The persistent per-workgroup accumulators (hist rows / minmax slots) fold to exactly the
    histogram / range of the one-shot entry points."""
    from ppq_amd import CUDA
    g = torch.Generator().manual_seed(13)
    batches = [torch.randn(n, generator=g) * 2 + 0.3 for n in (7, 4099, 1605632, 300000)]
    slots = torch.tensor([float('inf'), float('-inf')], device=DEV).repeat(CUDA.minmax_slots(), 1).contiguous()
    for b in batches: CUDA.MinMax_T_Slots(b.to(DEV), slots)
    mm = torch.tensor([float('inf'), float('-inf')], device=DEV)
    CUDA.MinMax_Slots_Finish(slots, mm)
    allv = torch.cat(batches)
    assert mm.cpu().tolist() == [allv.min().item(), allv.max().item()]
    for asym in (False, True):
        rows = torch.zeros(CUDA.hist_rows(), 2048, dtype=torch.int32, device=DEV)
        want = np.zeros(2048, np.int32)
        lo, hi = float(allv.min()) * 0.9, float(allv.max()) * 0.9
        hs = float(allv.abs().max()) / 2048 * 0.9
        for b in batches:
            if asym:
                CUDA.Histogram_Asymmetric_T_Rows(lo, hi, b.to(DEV), rows); O.hist_asym_t(b.numpy(), lo, hi, want)
            else:
                CUDA.Histogram_T_Rows(b.to(DEV), rows, hs); O.hist_sym_t(b.numpy(), hs, want)
        hist = torch.full([2048], 5, dtype=torch.int32, device=DEV)          # finish ADDS into hist
        CUDA.Histogram_Rows_Finish(rows, hist)
        assert np.array_equal(hist.cpu().numpy(), want + 5)


def test_sibling_prefetch_renders_the_same_and_never_stale():
    """observer.SIBLING_PREFETCH (on under install_plugins_into_ppq(observers=True), where PPQ's pass renders observer by observer):
    the first stand-alone render fetches ranges / searches histograms for every live observer.  Same scales as rendering each
    alone; an observer that observes AGAIN after a sibling's render must not be rendered from what was prefetched for it --
    in the range phase and in the histogram phase."""
    from ppq_amd import LinearQuantizationConfig, observer as obs_mod

    def build(kind, n):
        out = []
        for i in range(n):
            cfg = LinearQuantizationConfig(symmetrical=True, num_of_bits=8, quant_min=-128, quant_max=127)
            cfg.observer_algorithm = kind
            out.append(obs_mod.OBSERVER_TABLE[kind](watch_on=type('V', (), {'name': f'v{i}'})(), quant_cfg=cfg))
        return out
    g = torch.Generator().manual_seed(31)
    data = [[(torch.randn(40_000, generator=g) * (1 + i) + 0.1 * b).to(DEV) for b in range(3)] for i in range(6)]
    wide = (torch.randn(40_000, generator=g) * 50).to(DEV)

    def run(prefetch, kind):
        obs_mod.SIBLING_PREFETCH = prefetch
        try:
            obs = build(kind, 6)
            phases = 2 if kind in ('kl', 'mse') else 1
            for ph in range(phases):
                for ob, ds in zip(obs, data):
                    for d in ds: ob.observe(d)
                obs[0].render_quantization_config()          # with prefetch on: fetches / searches for obs[1..5] too
                obs[5].observe(wide)                         # .. but obs[5] observes once more before ITS render
                for ob in obs[1:]: ob.render_quantization_config()
            return [(float(ob._quant_cfg.scale), float(ob._quant_cfg.offset)) for ob in obs]
        finally:
            obs_mod.SIBLING_PREFETCH = False
    for kind in ('minmax', 'kl', 'mse'):
        a, b = run(False, kind), run(True, kind)
        assert a == b, (kind, a, b)
        assert a[5][0] > 2 * a[4][0]                         # the late, wide batch did reach observer 5


@pytest.mark.parametrize('mode', ['hip_graph', 'async', 'graph+async'])
def test_graph_and_async_modes_equal_eager(mode):
    """HIP-graph replay and side-stream observation change scheduling only: identical scales."""
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    g = torch.Generator().manual_seed(4)
    batches = [torch.rand(4, 3, 24, 24, generator=g).to(DEV) for _ in range(8)]

    def run(**kw):
        graph = harness.small_cnn_graph(seed=3)
        harness.quantize_graph(graph, 'kl', hist_bins=2048)
        ex = harness.TorchExecutor(graph, DEV)
        harness.ParameterQuantizePass().optimize(graph)
        p = RuntimeCalibrationPass(method='kl', **kw)
        p.optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
        torch.cuda.synchronize()
        return p, [float(c.scale) for op in graph.operations.values() for c, v in op.config_with_variable
                   if not v.is_parameter and c.state.value == 4]
    _, eager = run(use_hip_graph=False, async_observe=False)
    p, other = run(use_hip_graph='graph' in mode, async_observe='async' in mode)
    assert len(eager) == 6 and eager == other
    assert (p.graph_replays == 14) == ('graph' in mode)                   # 7 replays per phase


@pytest.mark.parametrize('batch', [1, 4])
def test_per_channel_activation_observers_under_hip_graph_capture(batch):
    """(batch 1: every activation is "one row per channel" and is queued into the multi-tensor launch; batch 4: direct launches.)
    ADVICE r4: per-channel min-max observers ride the multi-tensor launch (`ppqhip_minmax_c_multi`), whose job table now travels
    in the kernel arguments -- so the launch is legal inside a captured calibration forward.  Per-channel ACTIVATION configs
    (channel axis 1) + per-channel weights, `use_hip_graph=True`: the pass must really replay (no fallback) and leave the scales
    of the eager loop, bit for bit (min / max are order independent)."""
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    from ppq_amd.core import QuantizationPolicy, QuantizationProperty as QP
    g = torch.Generator().manual_seed(21)
    batches = [torch.rand(batch, 3, 24, 24, generator=g).to(DEV) * (1 + 0.1 * i) for i in range(8)]

    def run(**kw):
        graph = harness.small_cnn_graph(seed=3)
        harness.quantize_graph(graph, 'minmax')
        touched = 0
        for op in graph.operations.values():
            for c, v in op.config_with_variable:
                if v.is_parameter or c.dominated_by is not c or c.state.value != 1: continue          # INITIAL activation roots only
                c.policy = QuantizationPolicy(QP.LINEAR.value + QP.SYMMETRICAL.value + QP.PER_CHANNEL.value)
                c.channel_axis = 1
                touched += 1
        assert touched >= 3
        ex = harness.TorchExecutor(graph, DEV)
        harness.ParameterQuantizePass().optimize(graph)
        p = RuntimeCalibrationPass(method=None, **kw)
        p.optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
        torch.cuda.synchronize()
        out = []
        for op in graph.operations.values():
            for c, v in op.config_with_variable:
                if not v.is_parameter and c.dominated_by is c and c.state.value == 4 and c.channel_axis == 1:
                    out.append(c.scale.detach().cpu().clone())
        return p, out
    _, eager = run(use_hip_graph=False)
    p, replayed = run(use_hip_graph=True)
    assert p.graph_replays == 7                                           # one phase (min-max), 7 of the 8 batches replayed
    assert len(eager) == len(replayed) >= 3
    for a, b in zip(eager, replayed):
        assert a.numel() > 1 and torch.equal(a, b)


def test_auto_graph_mode_equals_eager():
    """use_hip_graph='auto' times one eager step per phase and captures only when launch-bound; either
    way the scales equal the eager loop's.  The small CNN at batch 4 is launch-bound, so with the
    minimum step count lowered both phases must choose the graph."""
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    g = torch.Generator().manual_seed(4)
    batches = [torch.rand(4, 3, 24, 24, generator=g).to(DEV) for _ in range(10)]

    def run(mode, min_steps):
        graph = harness.small_cnn_graph(seed=3)
        harness.quantize_graph(graph, 'kl', hist_bins=2048)
        ex = harness.TorchExecutor(graph, DEV)
        harness.ParameterQuantizePass().optimize(graph)
        p = RuntimeCalibrationPass(method='kl', use_hip_graph=mode)
        p.AUTO_GRAPH_MIN_STEPS = min_steps
        p.optimize(graph, dataloader=batches, executor=ex, calib_steps=10)
        torch.cuda.synchronize()
        return p, [float(c.scale) for op in graph.operations.values() for c, v in op.config_with_variable
                   if not v.is_parameter and c.state.value == 4]
    _, eager = run(False, 12)
    p, auto = run('auto', 4)
    assert eager == auto and len(p.graph_decisions) == 2
    assert all(d['graph'] for d in p.graph_decisions) and p.graph_replays == 16     # 8 replays per phase
    p, auto = run('auto', 12)                                                      # too few steps left: eager
    assert eager == auto and p.graph_replays == 0 and not any(d['graph'] for d in p.graph_decisions)


def test_lsq_autograd_functions():
    """CuLSQ_LT / CuLSQ_LC (training.py:17-90): forward == fake quant, backward == the _B kernels,
    checked against the torch formula of tests/test_cuda_kernel.py:67-78 (grad_x exact)."""
    from ppq_amd import LinearQuantizationConfig, QuantizationStates
    from ppq_amd.harness import Variable
    from ppq_amd.lsq import LSQDelegator
    from math import sqrt
    g = torch.Generator().manual_seed(8)
    t = (torch.rand(4, 6, 50, generator=g) * 50).to(DEV).requires_grad_(True)
    dy = torch.rand(4, 6, 50, generator=g).to(DEV)
    for per_channel in (False, True):
        cfg = LinearQuantizationConfig(symmetrical=False, quant_min=0, quant_max=255, channel_axis=1 if per_channel else None)
        cfg.scale = (torch.rand(6 if per_channel else 1, generator=g) * 0.5 + 0.05).to(DEV).requires_grad_(True)
        cfg.offset = torch.randint(0, 255, [6 if per_channel else 1], generator=g).float().to(DEV)
        cfg.state = QuantizationStates.ACTIVATED
        d = LSQDelegator(cfg, Variable('x', t, False))
        assert d.is_scale_trainable and d.is_offset_trainable and any(p is cfg.scale for p in d.trainable_tensors())
        y = d(t, cfg)
        y.backward(dy)
        view = [1, -1, 1] if per_channel else [1]
        s, o = cfg.scale.detach().view(view), cfg.offset.view(view)
        qt = torch.round(t.detach() / s) + o
        clipped = qt.clip(0, 255)
        assert torch.equal(y.detach(), (clipped - o) * s)
        dx = torch.where(clipped != qt, torch.zeros_like(dy), dy)
        assert torch.equal(t.grad, dx)
        ds = torch.where(clipped == qt, (((qt - o) * s) - t.detach()) * dy / s, torch.zeros_like(dy))
        ds = ds + torch.where(qt > 255, (255 - o) * dy, torch.zeros_like(dy)) + torch.where(qt < 0, (0 - o) * dy, torch.zeros_like(dy))
        if per_channel: want = ds.transpose(0, 1).flatten(1).sum(dim=-1) / sqrt(t.numel() * 255)          # linear.cu:402
        else: want = ds.sum().reshape(1) / sqrt(t.numel() * 255)                                             # linear.cu:299
        torch.testing.assert_close(cfg.scale.grad, want, rtol=1e-4, atol=1e-6)
        t.grad = None
        d.withdraw(); d.finalize()


def test_fp8_calibration_transformer_block():
    """BASELINE config 4 in miniature: FP8 E4M3 simulation of a transformer MLP block -- power-of-2
    scales from the 'floating' observer, fake-quant through FloatingQuantize_T/C, checked against the
    oracle on the tensors the executor produced."""
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    graph = harness.transformer_mlp_graph(seed=2)
    harness.quantize_graph_fp8(graph)
    ex = harness.TorchExecutor(graph, DEV)
    harness.ParameterQuantizePass().optimize(graph)
    g = torch.Generator().manual_seed(6)
    batches = [(torch.randn(8, 197, 64, generator=g) * 2).to(DEV) for _ in range(8)]
    RuntimeCalibrationPass().optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
    cands = {.0078125, .03125, .125, 1.0, 4.0, 16.0, 64.0}
    n_act = n_w = 0
    for op in graph.operations.values():
        for cfg, var in op.config_with_variable:
            if cfg.state.value != 4: continue
            assert all(float(s) in cands for s in cfg.scale.reshape(-1).tolist())
            if var.is_parameter:
                n_w += 1
                y = ex.quantize_function(var.value, cfg).cpu().numpy()
                want = O.fq_float_c(var.value.cpu().numpy(), cfg.scale.cpu().numpy(), cfg.offset.cpu().numpy(), 0)
                assert np.array_equal(y.view(np.uint32), want.view(np.uint32))
            else:
                n_act += 1
    assert (n_act, n_w) == (2, 2)
    # the quantised input of fc1 lies on the E4M3 grid scaled by its power-of-2 scale
    seen = {}

    class Spy:
        def pre_forward_hook(self, inputs, quant_inputs, quant_configs):
            seen['raw'], seen['q'], seen['cfg'] = inputs[0], quant_inputs[0], quant_configs[0]
            return quant_inputs
        def post_forward_hook(self, outputs, quant_outputs, quant_configs): return quant_outputs
    ex.forward(batches[0], hooks={'fc1': Spy()})
    cfg = seen['cfg']
    want = O.fq_float_t(seen['raw'].cpu().numpy(), cfg.scale.cpu().numpy().reshape(1), cfg.offset.cpu().numpy().reshape(1))
    assert np.array_equal(seen['q'].cpu().numpy().view(np.uint32), want.view(np.uint32))
    # baking the two FP8 weights: ONE ppqhip_fq_float_multi launch, the bits of the per-tensor function, same network output
    out_before = ex.forward(batches[0])[0].clone()
    weights = [(c, v) for op in graph.operations.values() for c, v in op.config_with_variable if v.is_parameter and c.state.value == 4]
    expected = {v.name: ex.quantize_function(v.value, c).clone() for c, v in weights}
    baking = harness.ParameterBakingPass()
    baking.optimize(graph)
    assert (baking.launches, baking.per_tensor) == (1, 0)
    for c, v in weights: assert c.state.value == 2 and torch.equal(v.value, expected[v.name])
    assert torch.equal(ex.forward(batches[0])[0], out_before)


@pytest.mark.parametrize('shape,axis', [((8, 64, 28, 28), 1), ((4, 512, 7, 7), 1), ((3, 5, 17), 1), ((32, 1000), 1),
                                         ((2, 6, 50, 50), 0), ((7, 33), -1), ((1, 16, 3), 2), ((64, 3, 224, 224), 1),
                                        ((3, 512, 56, 56), 1), ((2, 1024, 14, 14), 1), ((5, 600, 8, 12), 1)])       # C >= 2 per CU: the one-launch form (odd / even row counts, rows shorter and longer than a trip)
def test_channel_mean(CUDA, shape, axis):
    """ChannelMean == the per-channel torch.mean of BiasCorrectionPass.collect_bias
    (training.py:438-448); accumulated in double and in a fixed order, so the float32 result is the
    double-precision mean rounded to float32 (the float32 torch reduction is within 1e-6 relative of it) and two
    launches agree bit for bit."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(*shape, generator=g) * 3 + 0.5
    got = CUDA.ChannelMean(x.to(DEV), axis)
    ax = axis % x.ndim
    dims = tuple(i for i in range(x.ndim) if i != ax)
    want = x.double().mean(dim=dims).float()
    assert got.shape == want.shape
    assert torch.allclose(got.cpu(), want, rtol=1.2e-7, atol=1e-9)         # <= 1 float32 ulp of the double mean
    ref32 = x.mean(dim=dims)
    assert torch.allclose(got.cpu(), ref32, rtol=1e-6, atol=1e-6 * float(x.abs().mean()))
    assert torch.equal(CUDA.ChannelMean(x.to(DEV), axis), got)
    sums = torch.ones(x.shape[ax], dtype=torch.float64, device=DEV)              # accumulate semantics
    CUDA.ChannelSum(x.to(DEV), ax, sums)
    assert torch.allclose(sums.cpu(), x.double().sum(dim=dims) + 1, rtol=1e-12, atol=1e-9)


def test_bias_correction_pass():
    """BiasCorrectionPass (training.py:338-577) on an INT4-weight small CNN: every conv / gemm bias is
    shifted by mean(FP32 block output) - mean(quantised block output) per channel, which must (a) equal
    a plain-torch restatement of the same update for the first block and (b) never increase the block loss."""
    from ppq_amd import harness
    from ppq_amd.bias_correction import BiasCorrectionPass
    from ppq_amd.calibration import RuntimeCalibrationPass

    def build():
        graph = harness.small_cnn_graph(seed=5, width=16)
        harness.quantize_graph(graph, 'minmax')
        for op in graph.operations.values():
            for cfg, var in op.config_with_variable:
                if var.is_parameter and cfg.state.value == 1:
                    cfg.num_of_bits, cfg.quant_min, cfg.quant_max = 4, -8, 7
        ex = harness.TorchExecutor(graph, DEV)
        harness.ParameterQuantizePass().optimize(graph)
        return graph, ex
    g = torch.Generator().manual_seed(7)
    batches = [torch.rand(8, 3, 24, 24, generator=g).to(DEV) for _ in range(8)]
    graph, ex = build()
    RuntimeCalibrationPass().optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
    first = next(op for op in graph.topological_sort() if op.type == 'Conv')
    bias_before = {op.name: op.inputs[-1].value.clone() for op in graph.operations.values()
                   if op.type in ('Conv', 'Gemm') and len(op.inputs) == 3}
    # torch restatement of the first block's update (its input is the graph input)
    name = first.outputs[0].name
    first.dequantize()
    fp = [ex.partial_graph_forward([first], {first.inputs[0].name: b}, [name])[0].mean(dim=(0, 2, 3)) for b in batches]
    first.restore_quantize_state()
    qt = [ex.partial_graph_forward([first], {first.inputs[0].name: b}, [name])[0].mean(dim=(0, 2, 3)) for b in batches]
    want_err = torch.stack(fp).mean(0) - torch.stack(qt).mean(0)

    p = BiasCorrectionPass(steps=8)
    p.optimize(graph, dataloader=batches, executor=ex)
    assert len(p.report) == len(bias_before) >= 3
    assert all(post <= pre for _, pre, post in p.report)
    assert any(post < pre for _, pre, post in p.report)
    pre0, post0 = p.report[0][1:]
    got_err = first.inputs[-1].value - bias_before[first.name]
    if post0 < pre0:
        assert torch.allclose(got_err, want_err, rtol=1e-4, atol=1e-6)
    else:
        assert torch.equal(got_err, torch.zeros_like(got_err))
    # every config is back in its quantised state
    assert all(not getattr(op, '_dequantized', False) for op in graph.operations.values())
    # larger blocks (training.py:191-315): conv + relu chains become one block, the pass still never makes a block worse
    graph, ex = build()
    RuntimeCalibrationPass().optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
    p4 = BiasCorrectionPass(steps=8, block_size=4)
    p4.optimize(graph, dataloader=batches, executor=ex)
    assert 1 <= len(p4.report) <= len(bias_before) and all(post <= pre for _, pre, post in p4.report)


def test_multi_tensor_launches_equal_single(CUDA):
    """MinMax_T_Slots_Multi / Histogram_*_T_Rows_Multi: one launch over many tensors == one launch per
    tensor (bit-exact after the fold) and == the oracle; covers > 96 jobs (several launches), a
    1-element tensor, an unaligned view, accumulation over two rounds and a large tensor that needs
    more workgroups than there are rows."""
    g = torch.Generator().manual_seed(21)
    sizes = [1, 5, 1000, 4097, 65536, 300001, 1 << 20] * 14 + [40 * (1 << 20)]
    xs = [(torch.randn(n + 1, generator=g) * (1 + i % 5)).to(DEV)[(i % 2):][:n] for i, n in enumerate(sizes)]   # odd i: unaligned
    assert len(xs) == 99                                                  # > 96 jobs: two launches
    R, S, bins = CUDA.hist_rows(), CUDA.minmax_slots(), 2048
    # min / max -------------------------------------------------------------------------------
    seed = torch.tensor([float('inf'), float('-inf')], device=DEV)
    slots_a = [seed.repeat(S, 1).contiguous() for _ in xs]
    slots_b = [seed.repeat(S, 1).contiguous() for _ in xs]
    for rnd in range(2):
        ys = [x * (1 + rnd) for x in xs]
        CUDA.MinMax_T_Slots_Multi(ys, slots_a)
        for y, sl in zip(ys, slots_b): CUDA.MinMax_T_Slots(y, sl)
    for x, a, b in zip(xs, slots_a, slots_b):
        ra, rb = seed.clone(), seed.clone()
        CUDA.MinMax_Slots_Finish(a, ra); CUDA.MinMax_Slots_Finish(b, rb)
        assert torch.equal(ra, rb)
        assert float(ra[0]) == float((x * 2).min().clamp(max=float(x.min()))) and float(ra[1]) == float(torch.maximum(x.max(), (x * 2).max()))
    # symmetric + asymmetric histograms ----------------------------------------------------------
    scales = [float(x.abs().max()) / bins * 0.9 for x in xs]          # 0.9: some outliers get clipped
    los = [float(x.min()) * 0.8 for x in xs]; his = [float(x.max()) * 0.8 + 1e-3 for x in xs]
    for asym in (False, True):
        rows_a = [torch.zeros(R, bins, dtype=torch.int32, device=DEV) for _ in xs]
        rows_b = [torch.zeros(R, bins, dtype=torch.int32, device=DEV) for _ in xs]
        for rnd in range(2):
            if asym:
                CUDA.Histogram_Asymmetric_T_Rows_Multi(los, his, xs, rows_a)
                for x, r, lo, hi in zip(xs, rows_b, los, his): CUDA.Histogram_Asymmetric_T_Rows(lo, hi, x, r)
            else:
                CUDA.Histogram_T_Rows_Multi(xs, rows_a, scales)
                for x, r, s in zip(xs, rows_b, scales): CUDA.Histogram_T_Rows(x, r, s)
        for i, (x, a, b) in enumerate(zip(xs, rows_a, rows_b)):
            ha = torch.zeros(bins, dtype=torch.int32, device=DEV); hb = torch.zeros_like(ha)
            CUDA.Histogram_Rows_Finish(a, ha); CUDA.Histogram_Rows_Finish(b, hb)
            assert torch.equal(ha, hb), (asym, i)
            if x.numel() <= (1 << 20):
                want = np.zeros(bins, np.int32)
                xn = x.cpu().numpy()
                for _ in range(2):
                    if asym: O.hist_asym_t(xn, los[i], his[i], want, True)
                    else: O.hist_sym_t(xn, scales[i], want, True)
                assert np.array_equal(ha.cpu().numpy(), want), (asym, i)
    with pytest.raises(RuntimeError):
        CUDA.Histogram_T_Rows_Multi(xs[:2], [torch.zeros(R, bins, dtype=torch.int32, device=DEV)], scales[:2])


@pytest.mark.parametrize('method', ['kl', 'mse', 'minmax', 'percentile'])
def test_batched_observations_equal_per_tensor_launches(method):
    """RuntimeCalibrationPass(batch_observations=True) (one multi-tensor launch per forward and statistic)
    renders exactly the scales / offsets of the per-tensor launches."""
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    g = torch.Generator().manual_seed(4)
    batches = [torch.rand(4, 3, 24, 24, generator=g).to(DEV) - 0.3 for _ in range(8)]

    def run(batch):
        graph = harness.small_cnn_graph(seed=3)
        harness.quantize_graph(graph, method, hist_bins=2048 if method == 'kl' else None)
        ex = harness.TorchExecutor(graph, DEV)
        harness.ParameterQuantizePass().optimize(graph)
        p = RuntimeCalibrationPass(method=method, batch_observations=batch)
        p.optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
        torch.cuda.synchronize()
        return p, [(float(c.scale), float(c.offset)) for op in graph.operations.values() for c, v in op.config_with_variable
                   if not v.is_parameter and c.state.value == 4]
    p0, single = run(False)
    p1, multi = run(True)
    assert p0._queue is None and p1._queue is not None and p1._queue.launches >= 8
    assert len(single) == 6 and single == multi


@pytest.mark.parametrize('rounding', [0, 1, 4])
def test_linear_quantize_plan_equals_per_tensor_kernels(CUDA, rounding):
    """LinearQuantizePlan (one launch for many weights) == LinearQuantize_C / _T item by item, bit-exact:
    conv / gemm weights, elem_per_channel % 4 != 0 (scalar path), an unaligned view, per-tensor items,
    asymmetric and int4 ranges, > 128 items (several launches); in-place weight updates are picked up
    without rebuilding (the table holds pointers)."""
    from ppq_amd.ffi import LinearQuantizePlan
    g = torch.Generator().manual_seed(31)
    shapes = [(64, 3, 7, 7), (64, 64, 1, 1), (128, 64, 3, 3), (1000, 512), (7, 5, 3), (1, 1), (33, 1, 3, 3), (512, 2048, 1, 1)]
    items, want = [], []
    for rep in range(17):
        for i, shp in enumerate(shapes):
            w = (torch.randn(*shp, generator=g) * 0.2).to(DEV)
            if (rep + i) % 5 == 0:                                    # unaligned storage offset
                w = torch.cat([w.flatten(), w.flatten()[:1]])[1:].view(shp) if w.numel() > 1 else w
            per_channel = (rep + i) % 3 != 0
            C = shp[0] if per_channel else 1
            scale = (torch.rand(C, generator=g) * 0.01 + 1e-3).to(DEV)
            asym = (rep + i) % 2 == 0
            qmin, qmax = ((0, 255) if asym else (-128, 127)) if i % 4 else (-8, 7)
            offset = (torch.randint(0, 255, [C], generator=g).float() if asym else torch.zeros(C)).to(DEV)
            items.append((w, scale, offset, 0 if per_channel else None, qmin, qmax))
    assert len(items) == 136
    plan = LinearQuantizePlan(items, rounding=rounding)

    def reference():
        return [CUDA.LinearQuantize_C(w, s, o, ax, lo, hi, rounding) if ax is not None
                else CUDA.LinearQuantize_T(w, s, o, lo, hi, rounding) for w, s, o, ax, lo, hi in items]
    for got, ref, it in zip(plan.run(), reference(), items):
        assert got.shape == it[0].shape and torch.equal(got, ref)
    for w, s, *_ in items[::7]:                                       # in-place updates: no rebuild needed
        w.mul_(1.5); s.mul_(0.9)
    for got, ref in zip(plan.run(), reference()): assert torch.equal(got, ref)


@pytest.mark.parametrize('rounding', [0, 1])
def test_floating_quantize_plan_equals_per_tensor_kernels(CUDA, rounding):
    """FloatingQuantizePlan (ppqhip_fq_float_multi: one launch for all FP8 weights of a forward) == FloatingQuantize_C / _T
    item by item, bit-exact: E4M3 and E5M2 in one plan, power-of-two and odd scales (both arithmetic paths), per-channel and
    per-tensor items, elem_per_channel % 4 != 0, an unaligned view, > 128 items; in-place updates need no rebuild."""
    from ppq_amd.ffi import FloatingQuantizePlan
    g = torch.Generator().manual_seed(32)
    shapes = [(64, 3, 7, 7), (768, 768), (128, 64, 3, 3), (1000, 512), (7, 5, 3), (1, 1), (33, 1, 3, 3), (2304, 768)]
    pw = torch.tensor([.0078125, .03125, .125, 1.0, 4.0, 16.0, 64.0, 0.3])
    items = []
    for rep in range(17):
        for i, shp in enumerate(shapes):
            w = (torch.randn(*shp, generator=g) * (0.05 if i % 2 else 2.0)).to(DEV)
            if (rep + i) % 5 == 0:
                w = torch.cat([w.flatten(), w.flatten()[:1]])[1:].view(shp) if w.numel() > 1 else w
            per_channel = (rep + i) % 3 != 0
            C = shp[0] if per_channel else 1
            scale = pw[torch.randint(0, 8, [C], generator=g)].to(DEV)
            fmt = (4, 3, -448.0, 448.0) if (rep + i) % 2 else (5, 2, -57344.0, 57344.0)
            items.append((w, scale, torch.zeros(C, device=DEV), 0 if per_channel else None) + fmt)
    assert len(items) == 136
    plan = FloatingQuantizePlan(items, rounding=rounding)

    def reference():
        return [CUDA.FloatingQuantize_C(w, s, o, ax, e, m, lo, hi, rounding) if ax is not None
                else CUDA.FloatingQuantize_T(w, s, o, e, m, lo, hi, rounding) for w, s, o, ax, e, m, lo, hi in items]
    for got, ref, it in zip(plan.run(), reference(), items):
        assert got.shape == it[0].shape and torch.equal(got, ref)
    for w, s, *_ in items[::7]:
        w.mul_(1.5); s.mul_(2.0)
    for got, ref in zip(plan.run(), reference()): assert torch.equal(got, ref)


def test_fused_parameter_quantization_equals_per_weight_launches():
    """TorchExecutor(fuse_parameter_quantization=True): one launch for all weights per forward gives the
    forward outputs of the per-weight launches, follows re-rendered scales, and leaves delegated configs alone."""
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    g = torch.Generator().manual_seed(4)
    batches = [torch.rand(4, 3, 24, 24, generator=g).to(DEV) for _ in range(8)]

    def run(fuse):
        graph = harness.small_cnn_graph(seed=3)
        harness.quantize_graph(graph, 'kl', hist_bins=2048)
        ex = harness.TorchExecutor(graph, DEV)
        ex.fuse_parameter_quantization = fuse
        harness.ParameterQuantizePass().optimize(graph)
        RuntimeCalibrationPass(method='kl').optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
        return graph, ex, ex.forward(batches[0])[0].clone()
    _, ex0, y0 = run(False)
    graph, ex1, y1 = run(True)
    assert not ex0._plans and len(ex1._plans) == 1 and torch.equal(y0, y1)
    # a config edit (new scale tensor) is followed
    conv = next(op for op in graph.operations.values() if op.type == 'Conv')
    wcfg = conv.config.input_quantization_config[1]
    wcfg.scale = wcfg.scale * 2
    y2 = ex1.forward(batches[0])[0].clone()
    ex1.fuse_parameter_quantization = False
    assert torch.equal(ex1.forward(batches[0])[0], y2) and not torch.equal(y2, y1)
    # a delegated config is not part of the plan
    ex1.fuse_parameter_quantization = True
    ex1.register_quantize_delegate(wcfg, lambda t, c: t)
    ex1.forward(batches[0])
    assert (conv.inputs[1].name, id(wcfg)) not in ex1._fused


def test_multi_tensor_quantile_equals_single(CUDA):
    """Quantile_Multi (one launch per radix pass for many tensors) == Quantile per tensor == the oracle's
    order statistics (sort.cu:6-59 index rule), incl. > 64 jobs, 1-element and unaligned tensors,
    ReLU-style data (half the elements equal) and a tensor needing the workgroup cap."""
    g = torch.Generator().manual_seed(33)
    sizes = [1, 2, 7, 1000, 4097, 65536, 300001] * 10 + [40 * (1 << 20)]
    xs = []
    for i, n in enumerate(sizes):
        x = torch.randn(n + 1, generator=g) * (1 + i % 3)
        if i % 4 == 0: x = torch.relu(x)
        xs.append(x.to(DEV)[(i % 2):][:n])
    for q in (0.9999, 0.99, 0.5):
        multi = CUDA.Quantile_Multi(xs, q)
        for i, (x, got) in enumerate(zip(xs, multi)):
            single = CUDA.Quantile(x, q)
            assert torch.equal(got, single), (q, i)
            if x.numel() <= 300001:
                want = O.quantile_t(x.cpu().numpy(), q)
                assert np.array_equal(got.cpu().numpy(), np.asarray(want, np.float32)), (q, i)


def test_reuse_activations_equals_second_forward():
    """RuntimeCalibrationPass(reuse_activations=True) keeps the phase-1 activations in HBM and bins them in
    phase 2 instead of running the forward again: identical scales on an all-KL graph; on a graph where the
    phase-1 render activates an activation config (mixed minmax / kl) the kept tensors are dropped and the
    forward runs again (still identical); a small budget replays only the leading batches."""
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    g = torch.Generator().manual_seed(4)
    batches = [torch.rand(4, 3, 24, 24, generator=g).to(DEV) - 0.2 for _ in range(8)]

    def run(mixed, **kw):
        graph = harness.small_cnn_graph(seed=3)
        harness.quantize_graph(graph, 'kl', hist_bins=2048)
        if mixed:
            cfg = next(c for op in graph.topological_sort() if hasattr(op, 'config')
                       for c, v in op.config_with_variable if not v.is_parameter and c.state.value == 1)   # first INITIAL activation
            cfg.observer_algorithm = 'minmax'
        ex = harness.TorchExecutor(graph, DEV)
        harness.ParameterQuantizePass().optimize(graph)
        p = RuntimeCalibrationPass(method=None, **kw)
        p.optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
        torch.cuda.synchronize()
        return p, [float(c.scale) for op in graph.operations.values() for c, v in op.config_with_variable
                   if not v.is_parameter and c.state.value == 4]
    _, ref = run(False)
    p, got = run(False, reuse_activations=True)
    assert p.replayed_batches == 8 and got == ref and len(ref) == 6
    p, got = run(False, reuse_activations=True, reuse_budget_bytes=3 * 400_000)     # room for one or two batches
    assert 0 < p.replayed_batches < 8 and got == ref
    _, ref_mixed = run(True)
    p, got = run(True, reuse_activations=True)
    assert p.replayed_batches == 0 and got == ref_mixed and ref_mixed != ref


def test_vit_b16_fp8_calibration():
    """BASELINE config 4 at full size: ViT-B/16 topology (86.6 M parameters, 197 tokens), FP8 E4M3 simulation
    with the TRT_FP8 policy (inputs of Conv / Gemm / MatMul only; power-of-2 scales from the 'floating'
    observer).  Every activated config carries candidate scales, the fake-quantised operands the executor
    feeds to an attention MatMul and a per-channel weight equal the oracle's FP8 rounding bit for bit, and
    the FP8 network stays close to the FP32 one."""
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    graph = harness.vit_graph(seed=0)
    harness.quantize_graph_fp8(graph)
    ex = harness.TorchExecutor(graph, DEV)
    g = torch.Generator().manual_seed(6)
    batches = [torch.randn(2, 3, 224, 224, generator=g).to(DEV) for _ in range(8)]
    fp32_out = ex.forward(batches[0])[0].clone()              # every config still INITIAL: plain FP32 forward
    harness.ParameterQuantizePass().optimize(graph)
    RuntimeCalibrationPass().optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
    cands = {.0078125, .03125, .125, 1.0, 4.0, 16.0, 64.0}
    n_act = n_w = 0
    for op in graph.operations.values():
        for cfg, var in op.config_with_variable:
            if cfg.state.value != 4: continue
            assert all(float(s) in cands for s in cfg.scale.reshape(-1).tolist())
            n_w += var.is_parameter; n_act += not var.is_parameter
    assert (n_act, n_w) == (98, 50)
    seen = {}

    class Spy:
        def __init__(self, key): self.key = key
        def pre_forward_hook(self, inputs, quant_inputs, quant_configs):
            seen[self.key] = (inputs, quant_inputs, quant_configs)
            return quant_inputs
        def post_forward_hook(self, outputs, quant_outputs, quant_configs): return quant_outputs
    q_out = ex.forward(batches[0], hooks={'blk0_qk': Spy('qk'), 'blk5_qkv': Spy('qkv')})[0]
    raw, q, cfgs = seen['qk']                                  # attention scores: both operands are activations
    for r, y, c in zip(raw, q, cfgs):
        want = O.fq_float_t(r.contiguous().cpu().numpy(), c.scale.cpu().numpy().reshape(1), c.offset.cpu().numpy().reshape(1))
        assert np.array_equal(y.contiguous().cpu().numpy().view(np.uint32), want.view(np.uint32))
    raw, q, cfgs = seen['qkv']                                 # weight: per-channel FP8
    want = O.fq_float_c(raw[1].cpu().numpy(), cfgs[1].scale.cpu().numpy(), cfgs[1].offset.cpu().numpy(), 0)
    assert np.array_equal(q[1].cpu().numpy().view(np.uint32), want.view(np.uint32))
    cos = torch.nn.functional.cosine_similarity(q_out.flatten(), fp32_out.flatten(), dim=0)
    assert torch.isfinite(q_out).all() and float(cos) > 0.98, float(cos)


def _sharded_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        q.put((rank, _calibrate_config3(list(range(rank, 8, world)), dist.group.WORLD)))
    finally:
        dist.destroy_process_group()


def _calibrate_config3(batch_ids, group):
    """BASELINE config 3 in miniature: per-channel ASYMMETRIC int8 weights (minmax) + per-tensor asymmetric
    activations by MSE clipping search, on the batches `batch_ids` of a fixed 8-batch set."""
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    g = torch.Generator().manual_seed(12)
    batches = [torch.rand(4, 3, 24, 24, generator=g) - 0.4 for _ in range(8)]
    graph = harness.small_cnn_graph(seed=3)
    harness.quantize_graph(graph, 'mse', symmetrical=False, weight_symmetrical=False)
    ex = harness.TorchExecutor(graph, DEV)
    harness.ParameterQuantizePass().optimize(graph)
    p = RuntimeCalibrationPass(method='mse', process_group=group, check_steps=False)
    p.optimize(graph, dataloader=[batches[i].to(DEV) for i in batch_ids], executor=ex, calib_steps=len(batch_ids))
    torch.cuda.synchronize()
    return [(v.name, c.scale.reshape(-1).tolist(), c.offset.reshape(-1).tolist())
            for op in graph.operations.values() for c, v in op.config_with_variable if c.state.value == 4]


def test_sharded_mse_calibration_equals_union():
    """Two ranks (gloo, sharing this GPU), each calibrating half of the batches with the histogram /
    range all-reduce of distributed.merge_observers, render exactly the scales and offsets of one process
    over all batches -- integer SUM and float MIN/MAX are order independent."""
    import torch.multiprocessing as mp
    want = _calibrate_config3(list(range(8)), None)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs: p.join(timeout=60)
    assert len(want) >= 9 and any(len(s) > 1 for _, s, _ in want)           # activations + per-channel weights
    assert any(o != 0 for _, _, off in want for o in off)                   # asymmetric: non-zero offsets
    assert res[0] == want and res[1] == want


def test_channels_last_tensors_need_no_layout_copy(CUDA):
    """Dense channels-last tensors are streamed in storage order: per-tensor fake-quant returns the same
    values in the same memory format, statistics are those of the NCHW tensor, per-channel ops on axis 0
    (conv weights) too; a per-channel op on axis 1 still goes through the NCHW copy."""
    from ppq_amd.ffi import LinearQuantizePlan
    g = torch.Generator().manual_seed(41)
    x = (torch.randn(4, 24, 9, 7, generator=g) * 3).to(DEV)
    xcl = x.contiguous(memory_format=torch.channels_last)
    s, o = torch.tensor([0.05], device=DEV), torch.tensor([3.0], device=DEV)
    y, ycl = CUDA.LinearQuantize_T(x, s, o, 0, 255, 0), CUDA.LinearQuantize_T(xcl, s, o, 0, 255, 0)
    assert ycl.is_contiguous(memory_format=torch.channels_last) and torch.equal(y, ycl)
    h, hcl = torch.zeros(512, dtype=torch.int32, device=DEV), torch.zeros(512, dtype=torch.int32, device=DEV)
    CUDA.Histogram_T(x, h, 0.02); CUDA.Histogram_T(xcl, hcl, 0.02)
    assert torch.equal(h, hcl)
    mm = [torch.tensor([float('inf'), float('-inf')], device=DEV) for _ in range(2)]
    CUDA.MinMax_T(x, mm[0]); CUDA.MinMax_T(xcl, mm[1])
    assert torch.equal(mm[0], mm[1]) and torch.equal(CUDA.Quantile(x, 0.99), CUDA.Quantile(xcl, 0.99))
    w = (torch.randn(24, 4, 3, 3, generator=g) * 0.1).to(DEV)
    wcl = w.contiguous(memory_format=torch.channels_last)
    ws, wo = (torch.rand(24, generator=g) * 0.01 + 1e-3).to(DEV), torch.zeros(24, device=DEV)
    q, qcl = CUDA.LinearQuantize_C(w, ws, wo, 0, -128, 127, 0), CUDA.LinearQuantize_C(wcl, ws, wo, 0, -128, 127, 0)
    assert qcl.is_contiguous(memory_format=torch.channels_last) and torch.equal(q, qcl)
    got = LinearQuantizePlan([(wcl, ws, wo, 0, -128, 127), (xcl, s, o, None, 0, 255)]).run()
    assert got[0].is_contiguous(memory_format=torch.channels_last) and torch.equal(got[0], q) and torch.equal(got[1], y)
    cs, co = (torch.rand(24, generator=g) * 0.1 + 0.01).to(DEV), torch.zeros(24, device=DEV)
    assert torch.equal(CUDA.LinearQuantize_C(xcl, cs, co, 1, -128, 127, 0), CUDA.LinearQuantize_C(x, cs, co, 1, -128, 127, 0))


def test_channels_last_executor_calibrates_like_nchw():
    """TorchExecutor.use_channels_last(): same calibration (scales within the convolution algorithms'
    rounding), activations stay channels-last end to end."""
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    g = torch.Generator().manual_seed(4)
    batches = [torch.rand(4, 3, 24, 24, generator=g).to(DEV) for _ in range(8)]

    def run(cl):
        graph = harness.small_cnn_graph(seed=3)
        harness.quantize_graph(graph, 'kl', hist_bins=2048)
        ex = harness.TorchExecutor(graph, DEV)
        if cl: ex.use_channels_last()
        harness.ParameterQuantizePass().optimize(graph)
        RuntimeCalibrationPass(method='kl').optimize(graph, dataloader=batches, executor=ex, calib_steps=8)
        out = ex.forward(batches[0], output_names=['c1_relu_out'])[0]
        cfgs = [(c, v) for op in graph.operations.values() for c, v in op.config_with_variable if c.state.value == 4]
        return out, [float(c.scale) for c, v in cfgs if not v.is_parameter], [c.scale.clone() for c, v in cfgs if v.is_parameter]
    out0, act0, w0 = run(False)
    out1, act1, w1 = run(True)
    assert out1.is_contiguous(memory_format=torch.channels_last) and not out0.is_contiguous(memory_format=torch.channels_last)
    assert all(torch.equal(a, b) for a, b in zip(w0, w1))                    # weights: identical statistics
    assert act0 == pytest.approx(act1, rel=2e-2) and torch.allclose(out0, out1, rtol=1e-3, atol=1e-3)


def test_channelwise_kl_observer_equals_per_tensor_kl_on_each_channel(CUDA):
    """SURVEY 8f-4 extension: 'kl_channel' (per-channel two-phase KL; the reference's 'kl' refuses
    PER_CHANNEL) gives every channel exactly the scale the per-tensor 'kl' observer renders on that
    channel's slice; Histogram_C_Scales == Histogram_T per slice, bit for bit."""
    from ppq_amd.observer import TensorObserverFactroy, OBSERVER_TABLE, ChannelwiseKLObserver
    assert OBSERVER_TABLE['kl_channel'] is ChannelwiseKLObserver
    g = torch.Generator().manual_seed(51)
    C = 6
    data = [(torch.randn(3, C, 40, 30, generator=g) * torch.arange(1, C + 1).view(1, C, 1, 1) * (1 + 0.1 * i)) for i in range(4)]
    data = [torch.relu(d) if i % 2 else d for i, d in enumerate(data)]
    # kernel level
    scales = torch.tensor([0.01 * (c + 1) for c in range(C)])
    hist = torch.zeros(C, 2048, dtype=torch.int32, device=DEV)
    CUDA.Histogram_C_Scales(data[0].to(DEV), 1, hist, scales.to(DEV))
    for c in range(C):
        want = O.hist_sym_t(data[0][:, c].contiguous().numpy(), float(scales[c]), np.zeros(2048, np.int32), True)
        assert np.array_equal(hist[c].cpu().numpy(), want), c
    for shape, axis in (((5, 7, 9), 1), ((4, 3), 0), ((2, 3, 2000), 1)):               # short rows (global path), axis 0, long rows
        x = torch.randn(*shape, generator=g)
        Cx = shape[axis]
        sc = torch.rand(Cx, generator=g) * 0.01 + 0.002
        h = torch.zeros(Cx, 256, dtype=torch.int32, device=DEV)
        CUDA.Histogram_C_Scales(x.to(DEV), axis, h, sc.to(DEV), clip_outliers=False)
        for c in range(Cx):
            sl = x.select(axis, c).contiguous().numpy()
            assert np.array_equal(h[c].cpu().numpy(), O.hist_sym_t(sl, float(sc[c]), np.zeros(256, np.int32), False)), (shape, c)
    # observer level
    cfg = _cfg('kl_channel', per_channel_axis=1, bins=2048)
    ob = TensorObserverFactroy.build_observer('x', cfg)
    for _ in range(2):
        for d in data: ob.observe(d.to(DEV))
        ob.render_quantization_config()
    assert cfg.state.value == 4 and cfg.scale.shape == (C,) and float(cfg.offset.abs().sum()) == 0
    for c in range(C):
        ref_cfg = _cfg('kl', bins=2048)
        ref = TensorObserverFactroy.build_observer('x', ref_cfg)
        for _ in range(2):
            for d in data: ref.observe(d[:, c].contiguous().to(DEV))
            ref.render_quantization_config()
        assert float(ref_cfg.scale) == float(cfg.scale[c]), c
    with pytest.raises(ValueError):                                          # the reference's 'kl' still refuses per-channel
        bad = TensorObserverFactroy.build_observer('x', _cfg('kl', per_channel_axis=1, bins=2048))
        bad.observe(data[0].to(DEV)); bad.render_quantization_config()


def _run_bench(*flags, timeout=900):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), *flags], capture_output=True, text=True,
                       timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 ...` from a plain shell spawns two ranks (gloo, both on cuda:0 -- RCCL
    refuses duplicate devices), calibrates two shards and merges with ONE all-reduce per phase; the
    statistics are sums / minima over shards, so the scales do not depend on how many ranks there were:
    the checksum equals a 1-rank run over the same 2 x K batches... which needs the same seeds, so here
    only the contract is checked: n_gpus, merge records, value = all ranks' samples / max time."""
    small = ['--steps', '2', '--warmup', '1', '--batch', '4', '--repeats', '1', '--no-cpu-baseline', '--no-cpu-ops',
             '--pmc', '0', '--settle-ms', '0', '--variants', '0']
    out = _run_bench('--gpus', '2', '--backend', 'gloo', '--single-device', '1', *small)
    assert out['n_gpus'] == 2 and out['config']['rccl_ranks'] == 2 and out['config']['samples'] == 2 * 2 * 4
    merge = out['config']['merge']
    assert len(merge) == 2 and all(m['collectives'] == 1 and m['world_size'] == 2 for m in merge)
    assert merge[0]['min_f32_bytes'] == 72 * 2 * 4 and merge[1]['sum_int32_bytes'] == 72 * 2048 * 4
    assert abs(out['value'] - out['config']['samples'] / (out['ms_per_step'] * out['steps'] * 1e-3)) < 0.02 * out['value']
    one = _run_bench(*small)
    assert one['n_gpus'] == 1 and one['config']['merge'] == [] and one['roofline']['kernel'] in ('hist_sym_t', 'minmax_t', 'fq_linear_c')
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in one


def test_bench_strong_scaling_mode_two_ranks_on_one_gpu():
    """BASELINE config 3 is a FIXED job ("1024 samples sharded across 8"): `--total-samples S` derives the steps per rank from the
    job and the rank count and reports `scaling: strong`.  Two gloo ranks on this one GPU, 32 samples: 2 ranks x 4 steps x batch 4;
    the same job on one rank is 8 steps -- and, the statistics being sums / minima over samples, a job that is not a multiple of
    ranks x batch is refused rather than silently shrunk."""
    small = ['--warmup', '1', '--repeats', '1', '--no-cpu-baseline', '--no-cpu-ops', '--pmc', '0', '--settle-ms', '0', '--variants', '0', '--miopen-find', '0',
             '--workload', 'resnet50_cfg3', '--batch', '4', '--total-samples', '32']
    two = _run_bench('--gpus', '2', '--backend', 'gloo', '--single-device', '1', *small)
    assert two['scaling'] == 'strong' and two['n_gpus'] == 2 and two['steps'] == 4 and two['config']['samples'] == 32
    assert two['config']['total_samples'] == 32 and two['config']['merge_collectives_per_phase'] == 1
    assert two['config']['merge_phase1_ms'] > 0 and two['config']['merge_phase2_bytes'] == 72 * 2048 * 4
    one = _run_bench(*small)
    assert one['scaling'] == 'strong' and one['n_gpus'] == 1 and one['steps'] == 8 and one['config']['samples'] == 32
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--batch', '5', '--total-samples', '32'], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'not a multiple' in (r.stderr + r.stdout)


@pytest.mark.parametrize('workload', ['resnet50_cfg3', 'yolov6s_int4_lsq'])
def test_bench_two_ranks_on_one_gpu_configs_3_and_5(workload):
    """BASELINE configs 3 (ResNet-50, MSE search, asymmetric, histograms merged with an all-reduce) and 5 (YOLOv6-s-like INT4 LSQ,
    data-parallel gradients) through the real `bench.py --gpus 2` path: two gloo ranks sharing this one MI355X (RCCL refuses
    duplicate devices; the N-GPU run over RCCL is the driver's).  The contract of the 2-rank JSON line: n_gpus / rccl_ranks = 2,
    whole-job samples, the merge records of config 3 (a layout probe + ONE data collective per phase, the asymmetric range and
    the int32 histograms of all 72 observers in one flat buffer each), every block of config 5 trained on both ranks."""
    small = ['--warmup', '1', '--repeats', '1', '--no-cpu-baseline', '--no-cpu-ops', '--pmc', '0', '--settle-ms', '0', '--variants', '0', '--miopen-find', '0']
    if workload == 'resnet50_cfg3':
        out = _run_bench('--gpus', '2', '--backend', 'gloo', '--single-device', '1', '--workload', workload, '--steps', '2', '--batch', '4', *small)
        assert out['n_gpus'] == 2 and out['config']['rccl_ranks'] == 2 and out['config']['samples'] == 2 * 2 * 4
        merge = out['config']['merge']
        assert len(merge) == 2 and all(m['collectives'] == 1 and m['world_size'] == 2 for m in merge)
        assert merge[0]['min_f32_bytes'] == 72 * 2 * 4 and merge[1]['sum_int32_bytes'] == 72 * 2048 * 4
        assert 'mse' in out['metric'] and out['roofline']['kernel'] in ('hist_asym_t', 'minmax_t', 'fq_linear_c')
    else:
        out = _run_bench('--gpus', '2', '--backend', 'gloo', '--single-device', '1', '--workload', workload, '--steps', '2', '--batch', '2', *small)
        assert out['n_gpus'] == 2 and out['config']['rccl_ranks'] == 2
        assert out['config']['blocks'] == 27 and out['config']['optimizer_steps'] == 27 * 2
        assert out['config']['samples'] == 2 * 27 * 2 * 2                                   # ranks x blocks x steps x batch
        assert out['config']['execution']['graph_replays'] == 0                              # a gradient all-reduce per step: no HIP graph
    assert out['value'] > 0 and out['scaling'] == 'weak'


@pytest.mark.parametrize('sym', [True, False])
def test_channelwise_mse_observer_equals_per_tensor_mse_on_each_channel(CUDA, sym):
    """SURVEY 8f-4 extension: 'mse_channel' (per-channel two-phase histogram MSE; the reference's 'mse' raises on
    PER_CHANNEL, range.py:496-497) gives every channel exactly the (scale, offset) the per-tensor 'mse' observer
    renders on that channel's slice; Histogram_Asymmetric_C_Ranges == Histogram_Asymmetric_T per slice."""
    from ppq_amd.observer import TensorObserverFactroy, OBSERVER_TABLE, ChannelwiseMSEObserver
    assert OBSERVER_TABLE['mse_channel'] is ChannelwiseMSEObserver
    g = torch.Generator().manual_seed(61)
    C = 5
    data = [(torch.randn(3, C, 40, 30, generator=g) * torch.arange(1, C + 1).view(1, C, 1, 1) * (1 + 0.1 * i) + 0.3 * i) for i in range(4)]
    data = [torch.relu(d) if i % 2 else d for i, d in enumerate(data)]
    # kernel level: long rows (LDS path), short rows / axis 0 / axis last (global path), clipped and clamped
    for shape, axis in (((3, C, 40, 30), 1), ((5, 7, 9), 1), ((4, 3), 0), ((2, 3, 2000), 1), ((6, 50, 4), 2)):
        x = torch.randn(*shape, generator=g) * 2
        Cx = shape[axis]
        lo = torch.tensor([float(x.select(axis, c).min()) * 0.9 for c in range(Cx)])
        hi = torch.tensor([float(x.select(axis, c).max()) * 0.9 for c in range(Cx)])
        for clip in (True, False):
            h = torch.zeros(Cx, 256, dtype=torch.int32, device=DEV)
            CUDA.Histogram_Asymmetric_C_Ranges(x.to(DEV), axis, h, lo.to(DEV), hi.to(DEV), clip_outliers=clip)
            for c in range(Cx):
                sl = x.select(axis, c).contiguous().numpy()
                want = O.hist_asym_t(sl, float(lo[c]), float(hi[c]), np.zeros(256, np.int32), clip)
                assert np.array_equal(h[c].cpu().numpy(), want), (shape, axis, c, clip)
    # observer level
    cfg = _cfg('mse_channel', sym=sym, per_channel_axis=1)
    ob = TensorObserverFactroy.build_observer('x', cfg)
    for _ in range(2):
        for d in data: ob.observe(d.to(DEV))
        ob.render_quantization_config()
    assert cfg.state.value == 4 and cfg.scale.shape == (C,) and cfg.offset.shape == (C,)
    for c in range(C):
        ref_cfg = _cfg('mse', sym=sym)
        ref = TensorObserverFactroy.build_observer('x', ref_cfg)
        for _ in range(2):
            for d in data: ref.observe(d[:, c].contiguous().to(DEV))
            ref.render_quantization_config()
        assert float(ref_cfg.scale) == float(cfg.scale[c]) and float(ref_cfg.offset) == float(cfg.offset[c]), c
    with pytest.raises((ValueError, PermissionError)):                       # the reference's 'mse' still refuses per-channel (range.py:288-289, 496-497)
        bad = _cfg('mse', sym=sym, per_channel_axis=1)
        ob = TensorObserverFactroy.build_observer('x', bad)
        for _ in range(2):
            for d in data[:1]: ob.observe(d.to(DEV))
            ob.render_quantization_config()


def test_floating_observer_merged_squared_errors_pick_the_same_scale():
    """Data-parallel FP8 calibration: the per-candidate squared-error sums a rank contributes to the merge
    (DirectMSEObserver.reducible) select, after a merge, the same scale the stand-alone render selects from the same
    fetches -- per tensor and per channel; summing two ranks' contributions == observing both shards on one rank."""
    from ppq_amd.core import FloatingQuantizationConfig
    from ppq_amd.observer import DirectMSEObserver
    g = torch.Generator().manual_seed(8)
    var = type('V', (), {'name': 'x', 'is_parameter': False})()
    shards = [[(torch.randn(4, 16, 8, 8, generator=g) * s).to(DEV) for _ in range(3)] for s in (0.05, 30.0)]

    def observed(batches, seed):
        cfg = FloatingQuantizationConfig(calibration='floating')
        ob = DirectMSEObserver(var, cfg)
        torch.manual_seed(seed)
        for b in batches: ob.observe(b)
        return ob, cfg
    a, ca = observed(shards[0] + shards[1], 3)          # one rank sees everything
    a.render_quantization_config()
    b, cb = observed(shards[0] + shards[1], 3)          # the same fetches through the merged path
    bufs = b.reducible()
    assert [k for _, k in bufs] == ['sum', 'sum'] and bufs[0][0].dtype == torch.float64
    b.render_quantization_config()
    assert torch.equal(ca.scale, cb.scale)
    torch.manual_seed(3)                                # two ranks: contributions add up to the single-rank sums
    r0, _ = observed(shards[0], 3)
    r1, c1 = observed(shards[1], 4)
    s0, s1 = r0.reducible(), r1.reducible()
    for (x, _), (y, _) in zip(s0, s1): y += x            # what the SUM all-reduce does
    r1.render_quantization_config()
    assert float(c1.scale) in DirectMSEObserver.SCALE_CANDIDATES
    w = type('V', (), {'name': 'w', 'is_parameter': True})()
    cw = FloatingQuantizationConfig(calibration='floating', channel_axis=0)
    ow = DirectMSEObserver(w, cw); weight = (torch.randn(8, 4, 3, 3, generator=g) * torch.logspace(-3, 2, 8).view(8, 1, 1, 1)).to(DEV)
    ow.observe(weight); ow.render_quantization_config()
    cw2 = FloatingQuantizationConfig(calibration='floating', channel_axis=0)
    ow2 = DirectMSEObserver(w, cw2); ow2.observe(weight); ow2.reducible(); ow2.render_quantization_config()
    assert torch.equal(cw.scale, cw2.scale) and cw.scale.numel() == 8 and len(set(cw.scale.tolist())) > 1


def test_float_scale_search_kernel_and_batched_floating_render(CUDA):
    """ppqhip_float_scale_search: the squared fake-quant error of every row under every candidate == the sum over
    (FloatingQuantize_T(row, s) - row)^2 -- the float32 difference the reference forms, squared and summed in float64 (1e-12: only the
    summation order differs), for E4M3 / E5M2, power-of-two
    and odd candidates, HALF_EVEN and HALF_UP, rows of 1 .. 40000 values, > 128 items.  And render_observers() -- one search
    launch + one device->host copy for all 'floating' configs -- picks exactly the scales the stand-alone
    DirectMSEObserver.render_quantization_config() (the reference's literal arithmetic) picks."""
    from ppq_amd.core import FloatingQuantizationConfig
    from ppq_amd.observer import DirectMSEObserver, render_observers
    g = torch.Generator().manual_seed(12)
    items = []
    for k in range(140):
        rows, row_len = [(1, 40000), (7, 33), (64, 768), (3, 1), (16, 4097)][k % 5]
        v = (torch.randn(rows, row_len, generator=g) * float(10 ** ((k % 7) - 3))).to(DEV)
        items.append((v,) + ((4, 3, -448.0, 448.0) if k % 2 else (5, 2, -57344.0, 57344.0)))
    cands = [.0078125, .03125, .125, 1.0, 4.0, 16.0, 64.0, 0.3]
    zero = torch.zeros(1, device=DEV)
    for rounding in (0, 1):
        got = CUDA.FloatScaleSearch(items, cands, rounding)
        assert got.shape == (sum(it[0].shape[0] for it in items), len(cands)) and got.dtype == torch.float64
        starts = np.cumsum([0] + [it[0].shape[0] for it in items])
        for k in range(0, len(items), 9):
            (v, e, m, lo, hi), at = items[k], int(starts[k])
            for c, s in enumerate(cands):
                q = CUDA.FloatingQuantize_T(v, torch.full([1], s, device=DEV), zero, e, m, lo, hi, rounding)
                want = torch.sum(torch.square((q - v).double()), dim=-1)          # `qt - fp` in float32 as the reference forms it
                assert torch.allclose(got[at: at + v.shape[0], c], want, rtol=1e-12, atol=0), (e, m, s, rounding)
    # batched render == stand-alone render
    act = type('V', (), {'name': 'a', 'is_parameter': False})()
    par = type('V', (), {'name': 'w', 'is_parameter': True})()

    def build():
        gg = torch.Generator().manual_seed(77)
        obs = []
        for k in range(12):
            cfg = FloatingQuantizationConfig(calibration='floating')
            ob = DirectMSEObserver(act, cfg)
            torch.manual_seed(100 + k)
            for _ in range(3): ob.observe((torch.randn(2, 8, 16, 16, generator=gg) * float(4 ** (k % 6 - 3))).to(DEV))
            obs.append(ob)
        for k in range(6):
            cfg = FloatingQuantizationConfig(calibration='floating', channel_axis=0)
            ob = DirectMSEObserver(par, cfg)
            ob.observe((torch.randn(10, 4, 3, 3, generator=gg) * torch.logspace(-3, 2, 10).view(10, 1, 1, 1)).to(DEV))
            obs.append(ob)
        return obs
    alone, batched = build(), build()
    for ob in alone: ob.render_quantization_config()
    render_observers(batched)
    for a, b in zip(alone, batched):
        assert b._quant_cfg.state.value == 4
        assert torch.equal(a._quant_cfg.scale, b._quant_cfg.scale) and torch.equal(a._quant_cfg.offset, b._quant_cfg.offset)
    assert len({float(ob._quant_cfg.scale.flatten()[0]) for ob in batched}) >= 4          # the candidates were really exercised


@pytest.mark.parametrize('symmetrical', [True, False])
def test_isotone_observer_never_worse_than_minmax_at_keeping_the_argmax(symmetrical):
    """The property the reference's tests/test_isotone.py checks (10 000 random 10-class softmaxes per policy there, 1 500
    here, on the GPU through the HIP fake-quant kernel): after isotone calibration the quantised arg-max is wrong no
    more often than after min-max calibration of the same row."""
    from ppq_amd.core import LinearQuantizationConfig, QuantizationStates
    from ppq_amd.observer import OBSERVER_TABLE
    from ppq_amd.qfunction import PPQLinearQuantFunction
    cfg = LinearQuantizationConfig(symmetrical=symmetrical, quant_min=-128 if symmetrical else 0, quant_max=127 if symmetrical else 255,
                                   num_of_bits=8, calibration='isotone')
    var = type('V', (), {'name': 'TestVariable', 'is_parameter': False})()
    g = torch.Generator().manual_seed(1)
    rows = torch.sort(torch.softmax(torch.rand(1500, 10, generator=g), dim=-1), dim=-1)[0].to(DEV)
    for i in range(rows.shape[0]):
        value = rows[i: i + 1]
        errors = []
        for algo in ('isotone', 'minmax'):
            cfg.state = QuantizationStates.INITIAL
            ob = OBSERVER_TABLE[algo](var, cfg)
            ob.observe(value)
            ob.render_quantization_config()
            q = PPQLinearQuantFunction(value, cfg)
            errors.append(int(torch.sum(torch.argmax(value, dim=-1) != torch.argmax(q, dim=-1))))
        assert errors[0] <= errors[1], (i, errors, value, cfg.scale)


def test_isotone_calibration_pass_marks_softmax_outputs_and_calibrates_them():
    """optim/calibration.py:325-422: IsotoneCalibrationPass rewrites the Softmax output configs (INITIAL, 'Isotone', axis) AND,
    like the reference's (its optimize ends in super().optimize, :423), calibrates: the marked configs are rendered by the
    isotone observer inside this very pass, every other config by the algorithm it already had."""
    from ppq_amd import harness
    from ppq_amd.calibration import IsotoneCalibrationPass
    from ppq_amd.core import OBSERVER_ISOTONE_OBSERVER_AXIS
    graph = harness.vit_graph(seed=0, depth=1, dim=64, heads=2, mlp_dim=128, patch=16, num_classes=10)
    harness.quantize_graph(graph, 'minmax')
    ex = harness.TorchExecutor(graph, DEV)
    harness.ParameterQuantizePass().optimize(graph)
    g = torch.Generator().manual_seed(2)
    batches = [torch.randn(2, 3, 224, 224, generator=g).to(DEV) for _ in range(8)]
    seen = []
    from ppq_amd import observer as obs_mod
    orig = obs_mod.TorchIsotoneObserver.render_quantization_config

    def spy(self):
        seen.append(self)
        return orig(self)
    obs_mod.TorchIsotoneObserver.render_quantization_config = spy
    try:
        IsotoneCalibrationPass(verbose=False, calib_steps=8).optimize(graph, dataloader=batches, executor=ex)
    finally:
        obs_mod.TorchIsotoneObserver.render_quantization_config = orig
    marked = [op for op in graph.operations.values() if op.type == 'Softmax'
              and str(op.config.output_quantization_config[0].observer_algorithm).lower() == 'isotone']
    assert marked and all(OBSERVER_ISOTONE_OBSERVER_AXIS in op.config.output_quantization_config[0].detail for op in marked)
    assert len(seen) == len(marked)
    for op in marked:
        c = op.config.output_quantization_config[0]
        assert c.state.value == 4 and float(c.scale) > 0
    for op in graph.operations.values():                                   # nothing is left uncalibrated
        if hasattr(op, 'config'):
            for c, var in op.config_with_variable:
                assert c.state.value != 1, (op.name, var.name)
    assert torch.isfinite(ex.forward(batches[0])[0]).all()
    with pytest.raises(TypeError):
        IsotoneCalibrationPass(variables='x').optimize(graph, dataloader=batches, executor=ex)
    with pytest.raises(ValueError):
        IsotoneCalibrationPass(variables=['no such variable']).optimize(graph, dataloader=batches, executor=ex)



def test_lib_quant_stubs_observe_render_and_quantise():
    """ppq_amd.lib.TensorQuant / ParameterQuant (ppq/lib/quant.py:58-103): observe -> render -> forward gives exactly what the
    observer + PPQuantFunction give when driven by hand -- per-tensor activations (minmax, kl) and a per-channel weight -- and a
    delegator, once set, takes the call over."""
    import ppq_amd.lib as PFL
    from ppq_amd.qfunction import PPQuantFunction
    g = torch.Generator().manual_seed(3)
    batches = [torch.randn(4, 8, 16, 16, generator=g).to(DEV) for _ in range(4)]
    for algo in ('minmax', 'kl'):
        cfg_a, cfg_b = (PFL.LinearQuantizationConfig(calibration=algo) for _ in range(2))
        stub, ob = PFL.TensorQuant(cfg_a), PFL.Observer(cfg_b)
        for phase in range(2 if algo == 'kl' else 1):
            for b in batches: stub.observe(b); ob.observe(b)
            stub.render(); ob.render_quantization_config()
        assert cfg_a.state.value == 4 and torch.equal(cfg_a.scale, cfg_b.scale) and torch.equal(cfg_a.offset, cfg_b.offset)
        assert torch.equal(stub(batches[0]), PPQuantFunction(batches[0], cfg_b))
    w = torch.randn(16, 8, 3, 3, generator=g).to(DEV)
    cfg_w = PFL.LinearQuantizationConfig(channel_axis=0, calibration='minmax')
    pq = PFL.ParameterQuant(cfg_w, w)
    assert cfg_w.state.value == 4 and cfg_w.scale.shape == (16,)
    want = w.abs().amax(dim=(1, 2, 3)).double() * 2 / (cfg_w.quant_max - cfg_w.quant_min)
    assert torch.allclose(cfg_w.scale.double(), want, rtol=1e-6)
    assert torch.equal(pq(w), PPQuantFunction(w, cfg_w))
    pq.delegator = lambda t, c: t * 0
    assert float(pq(w).abs().max()) == 0.0
