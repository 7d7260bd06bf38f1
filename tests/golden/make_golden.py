"""Generate golden vectors by importing the REFERENCE (OpenPPL/ppq @ /root/reference) itself.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

Everything is produced by the reference's own PyTorch-CPU path (`USING_CUDA_KERNEL=False`),
i.e. the very code tests/test_cuda_kernel.py uses as the bit-exact yardstick for the CUDA
kernels.  Outputs (committed, all small):

    tests/golden/linear_fq.npz        per-tensor / per-channel linear fake-quant
    tests/golden/rounding.npz         ppq_tensor_round / ppq_numerical_round tables
    tests/golden/observers.npz        minmax / percentile / kl / mse observer results
    tests/golden/to_int.npz           PPQLinearQuant_toInt (quantise only, integer output; --to-int-only regenerates just this file)
    tests/golden/fp8_ref.npz          FP8 fake-quant + the 8 rounding modes, from the reference's common.cuh compiled on the host
                                      (--fp8-only regenerates just this file)
    tests/golden/dynamic.npz          PPQDyamicLinearQuantFunction, per tensor and per channel (--dynamic-only)
    tests/golden/kernels_ref.npz      histograms, quantile positions / picks, LSQ-backward and FP8-backward masks + scale gradients
                                      from the reference's OWN `__global__` kernel bodies (sort.cu / linear.cu / floating.cu) run on
                                      the host (oracle/_ref/libref_kernels.so; --kernels-only regenerates just this file)
    tests/golden/isotone.npz          TorchIsotoneObserver results (--isotone-only regenerates just this file)

Import shims (the container has no onnx and a newer protobuf/numpy than ppq expects):
stub `onnx*` modules, PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python, and a float() cast in
front of ppq_numerical_round (numpy>=2 no longer promotes np.float32/float to a Python float;
the cast is bit-identical to numpy-1 behaviour).
"""
import importlib.machinery
import os
import sys
from unittest.mock import MagicMock

os.environ['PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION'] = 'python'
sys.dont_write_bytecode = True
for _name in ['onnx', 'onnx.helper', 'onnx.numpy_helper', 'onnx.mapping', 'onnx.onnx_pb', 'onnx.checker',
              'onnx.external_data_helper', 'onnx.shape_inference', 'onnx.version_converter']:
    _m = MagicMock(); _m.__spec__ = importlib.machinery.ModuleSpec(_name, None); _m.__path__ = []
    sys.modules[_name] = _m
sys.path.insert(0, '/root/reference')

import numpy as np
import torch

import ppq  # noqa: E402  (the reference)
from ppq.core import (PPQ_CONFIG, QuantizationPolicy, QuantizationProperty, QuantizationStates,
                      RoundingPolicy, TensorQuantizationConfig)
from ppq.IR import Variable
from ppq.quantization.observer import range as ref_range
from ppq.quantization.observer import (TorchHistObserver, TorchMinMaxObserver, TorchMSEObserver,
                                       TorchPercentileObserver)
from ppq.quantization.qfunction import PPQLinearQuantFunction
from ppq.utils.round import ppq_numerical_round, ppq_round_to_power_of_2, ppq_tensor_round

assert PPQ_CONFIG.USING_CUDA_KERNEL is False
_orig_round = ref_range.ppq_numerical_round
ref_range.ppq_numerical_round = lambda v, *a, **k: _orig_round(float(v), *a, **k)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
P = QuantizationProperty


def tqc(per_channel=False, sym=True, qmin=-128, qmax=127, bits=8, axis=None, algo='minmax', pow2=False,
        rounding=RoundingPolicy.ROUND_HALF_EVEN, detail=None):
    pol = P.LINEAR + (P.PER_CHANNEL.value if per_channel else P.PER_TENSOR.value) \
        + (P.SYMMETRICAL.value if sym else P.ASYMMETRICAL.value) + (P.POWER_OF_2.value if pow2 else 0)
    return TensorQuantizationConfig(policy=QuantizationPolicy(pol), rounding=rounding, num_of_bits=bits,
                                    quant_min=qmin, quant_max=qmax, observer_algorithm=algo,
                                    channel_axis=axis, detail=detail)


def gen_linear():
    out = {}
    g = torch.Generator().manual_seed(20250117)
    cases = []
    # the shape / distribution matrix of tests/test_cuda_kernel.py:145-169 (big shapes shrunk to keep the fixture small)
    shapes_t = [[1, 1, 1, 1], [5, 12, 13, 4], [1, 7, 15, 41], [10, 12, 13, 4], [2, 7, 15, 41],
                [9, 7, 130, 1], [12, 4, 15, 3], [101, 7, 7, 1], [225, 1, 10, 4], [3, 10, 12, 47], [5, 42, 15, 3]]
    for i, shp in enumerate(shapes_t):
        for sym in (True, False):
            t = torch.rand(size=shp, generator=g) * 32
            s = torch.rand(size=[1], generator=g)
            o = torch.zeros(size=[1]) if sym else torch.randint(0, 255, size=[1], generator=g).float()
            cases.append(('t', t, s, o, None, 0, 255, RoundingPolicy.ROUND_HALF_EVEN))
    shapes_c = [([1, 1, 1, 1], 1), ([5, 12, 13, 4], 1), ([1, 7, 15, 41], 1), ([10, 12, 13, 4], 1),
                ([2, 7, 15, 41], 1), ([9, 7, 130, 1], 1), ([12, 4, 15, 3], 1), ([101, 7, 7, 1], 0),
                ([225, 1, 10, 4], 0), ([3, 10, 12, 47], 3), ([5, 42, 15, 3], 3), ([64, 3, 7, 7], 0),
                ([2, 97, 48], 2)]
    for shp, c in shapes_c:
        for sym in (True, False):
            t = torch.rand(size=shp, generator=g) * 32
            s = torch.rand(size=[shp[c]], generator=g)
            o = torch.zeros(size=[shp[c]]) if sym else torch.randint(0, 255, size=[shp[c]], generator=g).float()
            cases.append(('c', t, s, o, c, 0, 255, RoundingPolicy.ROUND_HALF_EVEN))
    # signed int8 / int4 on randn, all torch-supported rounding policies
    for pol in [RoundingPolicy.ROUND_HALF_EVEN, RoundingPolicy.ROUND_HALF_UP, RoundingPolicy.ROUND_HALF_DOWN,
                RoundingPolicy.ROUND_HALF_TOWARDS_ZERO, RoundingPolicy.ROUND_HALF_FAR_FORM_ZERO,
                RoundingPolicy.ROUND_UP]:
        for (qmin, qmax) in [(-128, 127), (-8, 7), (0, 15)]:
            t = torch.randn(size=[3, 8, 9, 5], generator=g) * 3
            s = torch.rand(size=[1], generator=g) * 0.1 + 0.01
            o = torch.zeros(size=[1]) if qmin < 0 else torch.tensor([float((qmax + 1) // 2)])
            cases.append(('t', t, s, o, None, qmin, qmax, pol))
            sc = torch.rand(size=[8], generator=g) * 0.1 + 0.01
            oc = torch.zeros(size=[8]) if qmin < 0 else torch.randint(0, qmax, size=[8], generator=g).float()
            cases.append(('c', t, sc, oc, 1, qmin, qmax, pol))
    # exact .5 ties (x = k/2 * s with s a power of two)
    t = (torch.arange(-600, 600).float() / 2.0) * 0.25
    cases.append(('t', t.reshape(2, 600), torch.tensor([0.25]), torch.tensor([0.0]), None, -128, 127,
                  RoundingPolicy.ROUND_HALF_EVEN))
    cases.append(('t', t.reshape(2, 600), torch.tensor([0.25]), torch.tensor([3.0]), None, 0, 255,
                  RoundingPolicy.ROUND_HALF_EVEN))

    out['n_cases'] = np.array(len(cases))
    for i, (kind, t, s, o, c, qmin, qmax, pol) in enumerate(cases):
        cfg = tqc(per_channel=(kind == 'c'), sym=True, qmin=qmin, qmax=qmax, axis=c, rounding=pol)
        cfg._scale, cfg._offset = s if kind == 'c' else s.squeeze(0), o if kind == 'c' else o.squeeze(0)
        cfg.state = QuantizationStates.ACTIVATED
        y = PPQLinearQuantFunction(t, cfg)
        out[f'{i}_kind'] = np.array(kind); out[f'{i}_x'] = t.numpy(); out[f'{i}_s'] = s.numpy()
        out[f'{i}_o'] = o.numpy(); out[f'{i}_axis'] = np.array(-1 if c is None else c)
        out[f'{i}_q'] = np.array([qmin, qmax]); out[f'{i}_rounding'] = np.array(pol.value)
        out[f'{i}_y'] = y.numpy()
    # BASELINE.md parity anchor (config 1)
    torch.manual_seed(0)
    x = torch.randn(1, 3, 224, 224)
    cfg = tqc(sym=True, qmin=-128, qmax=127)
    ob = TorchMinMaxObserver(Variable('x'), cfg); ob.observe(x); ob.render_quantization_config()
    y = PPQLinearQuantFunction(x, cfg)
    out['anchor_scale'] = np.array(float(cfg.scale.double()))
    out['anchor_scale_f32'] = cfg.scale.numpy()
    out['anchor_sum'] = np.array(float(y.double().sum()))
    out['anchor_maxerr'] = np.array(float((y - x).abs().max()))
    np.savez_compressed(os.path.join(HERE, 'linear_fq.npz'), **out)
    print('linear_fq.npz', len(cases), 'cases; anchor scale', out['anchor_scale'], 'sum', out['anchor_sum'])


def gen_rounding():
    out = {}
    grid = torch.cat([torch.arange(-40, 41).float() / 4.0,
                      torch.tensor([0.49999997, -0.49999997, 1e-8, -1e-8, 8388607.5, -8388607.5, 1e9, -1e9]),
                      torch.randn(200, generator=torch.Generator().manual_seed(7)) * 50])
    out['grid'] = grid.numpy()
    for pol in RoundingPolicy:
        if pol == RoundingPolicy.ROUND_TO_NEAR_INT: continue
        out[f'tensor_{pol.value}'] = ppq_tensor_round(grid, pol).numpy()
    vals = [1.5, 2.5, 0.5, -0.5, 1.1, 1.2, 1.3, -1.1, -1.2, -1.3, 3.5, -2.5, -1.5, 0.0, 7.49999, -7.5000001]
    out['num_values'] = np.array(vals)
    for pol in RoundingPolicy:
        out[f'num_{pol.value}'] = np.array([ppq_numerical_round(float(v), pol) for v in vals])
    p2 = [1.0, 1.2, 3.2, 0.26, 0.24, 0.5, 1e-8, 3e-5, 0.0078125, 100.0, -0.3]
    out['pow2_values'] = np.array(p2)
    out['pow2_up'] = np.array([ppq_round_to_power_of_2(v, RoundingPolicy.ROUND_UP) for v in p2])
    out['pow2_half_up'] = np.array([ppq_round_to_power_of_2(v, RoundingPolicy.ROUND_HALF_UP) for v in p2])
    np.savez_compressed(os.path.join(HERE, 'rounding.npz'), **out)
    print('rounding.npz')


def gen_observers():
    out = {}
    g = torch.Generator().manual_seed(42)
    batches = [torch.randn(2, 16, 14, 14, generator=g) * (1 + 0.1 * i) + 0.2 for i in range(4)]
    relu_batches = [torch.relu(b) for b in batches]
    out['batches'] = torch.stack(batches).numpy()

    def run(cls, cfg, data, two_phase=False, **kw):
        ob = cls(Variable('x'), cfg, **kw)
        for b in data: ob.observe(b)
        ob.render_quantization_config()
        if two_phase:
            for b in data: ob.observe(b)
            ob.render_quantization_config()
        return ob

    # --- minmax (per tensor / per channel, sym / asym, int8 / int4, pow2)
    k = 0
    for data_name, data in (('randn', batches), ('relu', relu_batches)):
        for per_channel in (False, True):
            for sym in (True, False):
                for (qmin, qmax, bits) in ((-128, 127, 8), (0, 255, 8), (-8, 7, 4)):
                    for pow2 in (False, True):
                        if pow2 and per_channel and not sym: pass
                        cfg = tqc(per_channel, sym, qmin, qmax, bits, axis=1 if per_channel else None, pow2=pow2)
                        run(TorchMinMaxObserver, cfg, data)
                        out[f'minmax_{k}_meta'] = np.array([data_name == 'relu', per_channel, sym, qmin, qmax, bits, pow2])
                        out[f'minmax_{k}_scale'] = cfg.scale.numpy(); out[f'minmax_{k}_offset'] = cfg.offset.numpy()
                        k += 1
    out['minmax_n'] = np.array(k)

    # --- percentile (CPU path: kthvalue, int(n*q) index)
    k = 0
    for data_name, data in (('randn', batches), ('relu', relu_batches)):
        for sym in (True, False):
            for pct in (0.9999, 0.999, 0.99):
                cfg = tqc(False, sym, -128 if sym else 0, 127 if sym else 255, algo='percentile',
                          detail={'OBSERVER_PERCENTILE_MANUL_OVERRIDE': pct})
                run(TorchPercentileObserver, cfg, data)
                out[f'pct_{k}_meta'] = np.array([data_name == 'relu', sym, pct])
                out[f'pct_{k}_scale'] = cfg.scale.numpy(); out[f'pct_{k}_offset'] = cfg.offset.numpy()
                k += 1
    out['pct_n'] = np.array(k)

    # --- KL (two phase, symmetric per tensor); keep the torch.histc histogram for the oracle
    k = 0
    for data_name, data in (('randn', batches), ('relu', relu_batches)):
        for bins in (2048, 4096, 512):
            for bits in (8, 4):
                for pow2 in (False, True):
                    qmin, qmax = (-128, 127) if bits == 8 else (-8, 7)
                    cfg = tqc(False, True, qmin, qmax, bits, algo='kl', pow2=pow2,
                              detail={'OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE': bins})
                    ob = TorchHistObserver(Variable('x'), cfg)
                    for b in data: ob.observe(b)
                    ob.render_quantization_config()
                    for b in data: ob.observe(b)
                    hist = ob._hist.clone()
                    ob.render_quantization_config()
                    out[f'kl_{k}_meta'] = np.array([data_name == 'relu', bins, bits, pow2])
                    out[f'kl_{k}_hist'] = hist.numpy().astype(np.int32)
                    out[f'kl_{k}_hist_scale'] = np.array(ob._hist_scale, np.float64)
                    out[f'kl_{k}_minmax'] = np.array([ob._min, ob._max], np.float64)
                    out[f'kl_{k}_scale'] = cfg.scale.numpy()
                    k += 1
    # a synthetic long-tailed histogram whose best range is NOT the last candidate
    hist = (np.exp(-np.arange(2048) / 90.0) * 50000).astype(np.int32); hist[1500:] = 0; hist[2047] = 3
    cfg = tqc(False, True, -128, 127, 8, algo='kl', detail={'OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE': 2048})
    ob = TorchHistObserver(Variable('x'), cfg)
    s, o = ob.hist_to_scale_offset(histogram=torch.tensor(hist), hist_bins=2048, hist_scale=0.01, config=cfg)
    out[f'kl_{k}_meta'] = np.array([0, 2048, 8, 0]); out[f'kl_{k}_hist'] = hist
    out[f'kl_{k}_hist_scale'] = np.array(0.01); out[f'kl_{k}_minmax'] = np.array([-20.48, 20.48])
    out[f'kl_{k}_scale'] = np.array(s, np.float32)
    k += 1
    out['kl_n'] = np.array(k)

    # --- MSE (two phase; pure-Python double loss loop == USING_CUDA_KERNEL False)
    k = 0
    for data_name, data in (('randn', batches), ('relu', relu_batches)):
        for sym in (True, False):
            for bins in (2048, 512):
                qmin, qmax = (-128, 127) if sym else (0, 255)
                cfg = tqc(False, sym, qmin, qmax, 8, algo='mse')
                ob = TorchMSEObserver(Variable('x'), cfg, bins=bins)
                for b in data: ob.observe(b)
                ob.render_quantization_config()
                for b in data: ob.observe(b)
                hist = ob._hist.clone()
                ob.render_quantization_config()
                out[f'mse_{k}_meta'] = np.array([data_name == 'relu', sym, bins, qmin, qmax])
                out[f'mse_{k}_hist'] = hist.numpy().astype(np.int32)
                out[f'mse_{k}_hist_scale'] = np.array(ob._hist_scale, np.float64)
                out[f'mse_{k}_minmax'] = np.array([ob._min, ob._max], np.float64)
                out[f'mse_{k}_scale'] = cfg.scale.numpy(); out[f'mse_{k}_offset'] = cfg.offset.numpy()
                # a few raw loss values of the Python loop
                hl = hist.tolist()
                probes = [(0, 1, 256), (8, 2, 520), (0, bins // 256 + 1, 256 * (bins // 256 + 1)), (16, 1, 272)]
                out[f'mse_{k}_probes'] = np.array(probes)
                out[f'mse_{k}_probe_loss'] = np.array([ob.compute_mse_loss(hl, s, st, e) for (s, st, e) in probes])
                k += 1
    out['mse_n'] = np.array(k)
    np.savez_compressed(os.path.join(HERE, 'observers.npz'), **out)
    print('observers.npz', {n: int(out[n]) for n in ('minmax_n', 'pct_n', 'kl_n', 'mse_n')})


def gen_observers_cuda_rule():
    """KL / MSE scales the reference renders when ITS CUDA kernels collect the histogram.

    The reference's CPU path bins with torch.histc (x == max lands in the last bin), its CUDA kernels drop
    that element (sort.cu:84-86) and score MSE candidates with the float loss of csrc/cpu/hist_mse.cc
    instead of the Python double loop (range.py:423-425).  On small tensors one count can flip the arg-min,
    so round 1 could only assert "most" scales against the CPU goldens.  Here the reference's OWN
    `hist_to_scale_offset` / `render_quantization_config` run on a histogram produced by the CUDA bin rule
    (restated from sort.cu:75-139 in oracle/ppq_oracle.c and pinned to the reference's own tolerance) and,
    for MSE, with `USING_CUDA_KERNEL = True` and the reference's own hist_mse.cc compiled where it lies
    (oracle/_ref) installed as `compute_mse_loss` -- exactly the code a CUDA run of the reference executes
    on the host.  tests/test_gpu_calibration.py asserts 100 % agreement with these."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import ppq_oracle as O
    from ppq.core.ffi import CUDA_COMPLIER
    out = {}
    g = torch.Generator().manual_seed(42)
    batches = [torch.randn(2, 16, 14, 14, generator=g) * (1 + 0.1 * i) + 0.2 for i in range(4)]
    relu_batches = [torch.relu(b) for b in batches]

    class HostExtension:                     # only the host helper is reachable with CPU tensors
        @ staticmethod
        def compute_mse_loss(hist, start, step, end): return O.ref_mse_loss(hist, start, step, end)
    CUDA_COMPLIER.__CUDA_EXTENTION__ = HostExtension()

    def cuda_hist(ob, data, sym, bins):
        h = np.zeros(bins, np.int32)
        for b in data:
            if sym: O.hist_sym_t(b.numpy(), np.float32(ob._hist_scale), h)
            else: O.hist_asym_t(b.numpy(), np.float32(ob._min), np.float32(ob._max), h)
        return torch.from_numpy(h)

    k = 0
    for data_name, data in (('randn', batches), ('relu', relu_batches)):
        for bins in (2048, 4096, 512):
            for bits in (8, 4):
                for pow2 in (False, True):
                    qmin, qmax = (-128, 127) if bits == 8 else (-8, 7)
                    cfg = tqc(False, True, qmin, qmax, bits, algo='kl', pow2=pow2,
                              detail={'OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE': bins})
                    ob = TorchHistObserver(Variable('x'), cfg)
                    for b in data: ob.observe(b)
                    ob.render_quantization_config()
                    ob._hist = cuda_hist(ob, data, True, bins)
                    ob.render_quantization_config()
                    out[f'kl_{k}_meta'] = np.array([data_name == 'relu', bins, bits, pow2])
                    out[f'kl_{k}_scale'] = cfg.scale.numpy()
                    k += 1
    out['kl_n'] = np.array(k)
    k = 0
    PPQ_CONFIG.USING_CUDA_KERNEL = True
    try:
        for data_name, data in (('randn', batches), ('relu', relu_batches)):
            for sym in (True, False):
                for bins in (2048, 512):
                    qmin, qmax = (-128, 127) if sym else (0, 255)
                    cfg = tqc(False, sym, qmin, qmax, 8, algo='mse')
                    ob = TorchMSEObserver(Variable('x'), cfg, bins=bins)
                    for b in data: ob.observe(b)
                    ob.render_quantization_config()
                    ob._hist = cuda_hist(ob, data, sym, bins)
                    ob.render_quantization_config()
                    out[f'mse_{k}_meta'] = np.array([data_name == 'relu', sym, bins, qmin, qmax])
                    out[f'mse_{k}_scale'] = cfg.scale.numpy(); out[f'mse_{k}_offset'] = cfg.offset.numpy()
                    k += 1
    finally:
        PPQ_CONFIG.USING_CUDA_KERNEL = False
    out['mse_n'] = np.array(k)
    np.savez_compressed(os.path.join(HERE, 'observers_cuda_rule.npz'), **out)
    print('observers_cuda_rule.npz', {n: int(out[n]) for n in ('kl_n', 'mse_n')})


def gen_fp8_ref():
    """FP8 outputs PRODUCED BY THE REFERENCE'S OWN SOURCE: ppq/csrc/cuda/common.cuh (QuantizeScalarFloating, _round2int,
    DequantizeScalar) compiled as host C++ where it lies (oracle/_ref/libref_common.so, `make -C oracle ref`), on the
    structured sweep of oracle/ref_common.py::sweep_bits (+ 8192 random bit patterns, seed 0).  Inputs are regenerated by
    the tests from the same function; only the outputs (uint32 bit patterns) are stored."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import ref_common as R
    assert R.available(), 'run `make -C oracle ref` first'
    out = {}
    for key, fmt, scale, offset, clip, rounding in R.fp8_ref_cases():
        E, M, c = R.FORMATS[fmt]
        if clip is not None: c = clip
        x = R.sweep_bits(M, n_random=8192, seed=0)
        y = R.fq_float_t(x, [scale], [offset], E, M, -c, c, rounding)
        out[key] = y.view(np.uint32)
    # _round2int, all 8 modes, on the values where the modes differ
    v = np.concatenate([np.arange(-64, 65) / 8.0, np.array([0.49999997, -0.49999997, 8388607.5, -8388607.5, 1e9, -1e9, 2.5e-7]),
                        np.random.default_rng(3).standard_normal(4096) * 300]).astype(np.float32)
    out['round_values'] = v
    out['round_results'] = np.array([[R.round2int(float(a), r) for a in v] for r in range(8)], np.int32)
    np.savez_compressed(os.path.join(HERE, 'fp8_ref.npz'), **out)
    print('fp8_ref.npz', len(out), 'arrays,', os.path.getsize(os.path.join(HERE, 'fp8_ref.npz')), 'bytes')


def gen_kernels_ref():
    """Outputs PRODUCED BY THE REFERENCE'S OWN KERNEL BODIES (oracle/ref_kernels.py: the `__global__` functions of sort.cu,
    linear.cu, floating.cu extracted at build time from where they lie and executed thread by thread on the host), on the
    seeded inputs of tests/golden/kernel_ref_cases.py.  Only outputs are stored: histograms, the two positions `_Quantile_T`
    reads per (n, q), its picks on small tensors, and for the backward kernels the bit mask `grad_x != 0`, grad_s, and the
    double-precision sum of the per-thread terms handed to the block reduction."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import ref_kernels as K
    import kernel_ref_cases as cases
    out = {}
    for key, kind, v, prm in cases.hist_cases():
        if kind == 'sym':
            out['hist_' + key] = K.hist_sym_t(v, prm[0], np.zeros(cases.BINS, np.int32), prm[1])
        elif kind == 'asym':
            out['hist_' + key] = K.hist_asym_t(v, prm[0], prm[1], np.zeros(cases.BINS, np.int32), prm[2])
        else:
            shape, axis, hs, clip = prm
            out['hist_' + key] = K.hist_sym_c(v, axis, hs, np.zeros(shape[axis] * 128, np.int32), clip)
    out['qpos'] = np.array([[K.quantile_positions(n, float(np.float32(q))) for q in cases.QUANTILE_QS] for n in cases.QUANTILE_NS], np.int64)
    for key, v, q in cases.quantile_arrays():
        out[key] = K.quantile_t(v, q)
    for key, x, dy, s, o, axis, qmin, qmax, r in cases.lsq_cases():
        if axis is None: gx, gs, part = K.fq_linear_t_bwd(x, s, o, dy, qmin, qmax, r, with_partials=True)
        else: gx, gs, part = K.fq_linear_c_bwd(x, s, o, dy, axis, qmin, qmax, r, with_partials=True)
        assert np.all((gx == 0) | (gx == dy))
        out[key + '_mask'] = np.packbits(gx.reshape(-1) != 0)
        out[key + '_gs'] = gs
        out[key + '_psum'] = part.astype(np.float64).sum(axis=None if axis is None else 1)
    for key, x, dy, s, o, axis, E, M, c in cases.fp8_bwd_cases():
        gx, gs = K.fq_float_c_bwd(x, s, o, dy, axis, E, M, -c, c, 0)
        assert np.all((gx == 0) | (gx == dy))
        out[key + '_mask'] = np.packbits(gx.reshape(-1) != 0)
        out[key + '_gs'] = gs
    np.savez_compressed(os.path.join(HERE, 'kernels_ref.npz'), **out)
    print('kernels_ref.npz', len(out), 'arrays,', os.path.getsize(os.path.join(HERE, 'kernels_ref.npz')), 'bytes')


from to_int_cases import to_int_cases, to_int_inputs  # noqa: E402  (shared with the tests; imports nothing of the reference)


def gen_to_int():
    from ppq.quantization.qfunction.linear import PPQLinearQuant_toInt
    out = {}
    for key, shape, axis, sym, bits, qmin, qmax, r in to_int_cases():
        x, scale, offset = to_int_inputs(key, shape, axis)
        cfg = tqc(per_channel=axis is not None, sym=sym, qmin=qmin, qmax=qmax, bits=bits, axis=axis, rounding=RoundingPolicy(r))
        cfg.scale, cfg.offset, cfg.state = scale, offset, QuantizationStates.ACTIVATED
        y = PPQLinearQuant_toInt(x, cfg)
        out[key] = y.numpy()
    np.savez_compressed(os.path.join(HERE, 'to_int.npz'), **out)
    print('to_int.npz', len(out), 'arrays,', os.path.getsize(os.path.join(HERE, 'to_int.npz')), 'bytes')


from dynamic_cases import dynamic_cases, dynamic_input  # noqa: E402


def gen_dynamic():
    """PPQDyamicLinearQuantFunction of the reference (qfunction/linear.py:99-198: min / max of THIS tensor, per tensor or per
    channel -> minmax_to_scale_offset -> torch fake quant) on the seeded cases of dynamic_cases.py."""
    from ppq.quantization.qfunction.linear import PPQDyamicLinearQuantFunction
    import io, contextlib
    out = {}
    for key, shape, axis, sym, qmin, qmax, pow2 in dynamic_cases():
        x = dynamic_input(key, shape, axis)
        cfg = tqc(per_channel=axis is not None, sym=sym, qmin=qmin, qmax=qmax, bits=8, axis=axis, pow2=pow2)
        cfg._policy = QuantizationPolicy(cfg.policy._policy + P.DYNAMIC.value)
        cfg.state = QuantizationStates.ACTIVATED
        with contextlib.redirect_stdout(io.StringIO()):             # the reference prints scale / offset (linear.py:120)
            y = PPQDyamicLinearQuantFunction(x, cfg)
        out[key] = y.numpy()
    np.savez_compressed(os.path.join(HERE, 'dynamic.npz'), **out)
    print('dynamic.npz', len(out), 'arrays,', os.path.getsize(os.path.join(HERE, 'dynamic.npz')), 'bytes')
def gen_isotone():
    """observer/order.py: TorchIsotoneObserver on classification-like outputs (the reference's tests/test_isotone.py draws
    softmax rows): multi-batch, single row, 3-D with the class axis last, an axis in the middle, and rows without any
    candidate (min-max fall-back); symmetric and asymmetric."""
    from ppq.core import OBSERVER_ISOTONE_OBSERVER_AXIS
    from ppq.quantization.observer import TorchIsotoneObserver
    g = torch.Generator().manual_seed(20240)
    cases = [
        ([torch.softmax(torch.randn(64, 10, generator=g) * 3, dim=-1) for _ in range(4)], -1),
        ([torch.softmax(torch.rand(1, 10, generator=g), dim=-1)], -1),
        ([torch.softmax(torch.randn(2, 7, 5, generator=g), dim=-1)], -1),
        ([torch.softmax(torch.randn(6, 12, generator=g) * 2, dim=-1) for _ in range(3)], 1),
        ([torch.randn(32, 100, generator=g)], -1),                       # logits, negative entries
        ([torch.full([3, 4], 0.25)], -1),                                 # no candidate -> min-max fall-back
    ]
    out, k = {}, 0
    for batches, axis in cases:
        for sym in (True, False):
            cfg = tqc(sym=sym, qmin=-128 if sym else 0, qmax=127 if sym else 255, algo='isotone',
                      detail={OBSERVER_ISOTONE_OBSERVER_AXIS: axis})
            cfg.state = QuantizationStates.INITIAL
            ob = TorchIsotoneObserver(Variable(name='x'), cfg)
            for b in batches: ob.observe(b)
            ob.render_quantization_config()
            out[f'iso_{k}_n'] = np.array(len(batches))
            for i, b in enumerate(batches): out[f'iso_{k}_x{i}'] = b.numpy()
            out[f'iso_{k}_meta'] = np.array([int(sym), axis, cfg.quant_min, cfg.quant_max])
            out[f'iso_{k}_scale'] = cfg.scale.reshape(-1).numpy().astype(np.float32)
            out[f'iso_{k}_offset'] = cfg.offset.reshape(-1).numpy().astype(np.float32)
            k += 1
    out['iso_n'] = np.array(k)
    np.savez_compressed(os.path.join(HERE, 'isotone.npz'), **out)
    print('isotone.npz', k)


if __name__ == '__main__':
    torch.set_num_threads(8)
    if '--to-int-only' in sys.argv:
        gen_to_int(); sys.exit(0)
    if '--fp8-only' in sys.argv:
        gen_fp8_ref(); sys.exit(0)
    if '--dynamic-only' in sys.argv:
        gen_dynamic(); sys.exit(0)
    if '--kernels-only' in sys.argv:
        gen_kernels_ref(); sys.exit(0)
    if '--isotone-only' in sys.argv:
        gen_isotone(); sys.exit(0)
    if '--cuda-rule-only' not in sys.argv:
        gen_linear()
        gen_rounding()
        gen_observers()
    gen_observers_cuda_rule()
    gen_fp8_ref()
    gen_kernels_ref()
    gen_to_int()
    gen_dynamic()
    gen_isotone()
