"""Drive the REAL reference (OpenPPL/ppq, unmodified) -- TEST / BASELINE INFRASTRUCTURE ONLY.

The reference is Python and cannot be compiled into ``oracle/_ref``; where a copy of it is importable
(``/root/reference`` in the build container, or the git-ignored staging directory
``oracle/_ref/ppq_stage`` that ``tools/stage_reference.py`` fills for a GPU run) this module imports
it with the three shims of SURVEY.md section 8c (stub ``onnx``, PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION,
``float()`` in front of ``ppq_numerical_round`` for numpy >= 2) and runs ITS graph IR, ITS TensorRT
quantizer, ITS ``TorchExecutor`` and ITS ``RuntimeCalibrationPass`` on a topology converted from a
``ppq_amd.harness`` graph (same names, same weights):

* on the CPU with ``USING_CUDA_KERNEL = False``: the reference's own numbers (bench.py ``cpu_baseline``
  kind "reference", tests pinning oracle/torch_cpu_path.py);
* on the GPU after ``ppq_amd.install_into_ppq()``: the unmodified reference executor running on the HIP
  kernels (tests/test_gpu_reference.py, tools/run_reference_tests.py).

Nothing under ``ppq_amd/`` imports this file.
"""
import importlib.machinery
import os
import sys
import time
from typing import Dict, List, Optional
from unittest.mock import MagicMock

import torch

_LOADED: Optional[str] = None


def find_reference() -> Optional[str]:
    """Path of an importable reference checkout: the staged copy first, then /root/reference."""
    here = os.path.dirname(os.path.abspath(__file__))
    for path in (os.path.join(here, '_ref', 'ppq_stage'), '/root/reference'):
        if os.path.isdir(os.path.join(path, 'ppq')): return path
    return None


def load(path: Optional[str] = None) -> str:
    """Import the reference from `path` (default: find_reference()).  Idempotent."""
    global _LOADED
    if _LOADED is not None: return _LOADED
    path = path or find_reference()
    if path is None: raise ImportError('no reference checkout is importable (stage one with tools/stage_reference.py)')
    os.environ.setdefault('PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION', 'python')
    sys.dont_write_bytecode = True
    try:
        import onnx  # noqa: F401
    except ImportError:
        for name in ['onnx', 'onnx.helper', 'onnx.numpy_helper', 'onnx.mapping', 'onnx.onnx_pb', 'onnx.checker',
                     'onnx.external_data_helper', 'onnx.shape_inference', 'onnx.version_converter']:
            m = MagicMock(); m.__spec__ = importlib.machinery.ModuleSpec(name, None); m.__path__ = []
            sys.modules[name] = m
    if path not in sys.path: sys.path.insert(0, path)
    import ppq  # noqa: F401  (the reference)
    from ppq.quantization.observer import range as ref_range
    if not getattr(ref_range, '_ppq_amd_numpy2_shim', False):
        orig = ref_range.ppq_numerical_round
        ref_range.ppq_numerical_round = lambda v, *a, **k: orig(float(v), *a, **k)     # numpy >= 2: range.py:59 / round.py:76
        ref_range._ppq_amd_numpy2_shim = True
    _LOADED = path
    return path


def _pair(v):
    return [int(v), int(v)] if isinstance(v, int) else [int(a) for a in v]


def to_reference_graph(hgraph):
    """ppq_amd.harness.BaseGraph (plain or quantised; only names, attributes and parameter values are read)
    -> ppq.BaseGraph built with the reference's own graph API (tests/test_persus.py:6-27 pattern)."""
    load()
    from ppq import BaseGraph
    from ppq.core import NetworkFramework
    g = BaseGraph(name=hgraph.name, built_from=NetworkFramework.ONNX)
    made = {}

    def var(v):
        if v.name not in made:
            value = v.value.detach().clone().cpu() if (v.is_parameter and v.value is not None) else None
            made[v.name] = g.create_variable(name=v.name, value=value, is_parameter=v.is_parameter)
        return made[v.name]

    for op in hgraph.operations.values():
        a = dict(op.attributes)
        if op.type == 'Conv':
            k = op.inputs[1].value.shape[2:]
            p = _pair(a.get('pads', 0))
            attrs = {'kernel_shape': [int(d) for d in k], 'strides': _pair(a.get('strides', 1)), 'pads': p + p,
                     'dilations': [1, 1], 'group': int(a.get('group', 1))}
        elif op.type == 'MaxPool':
            p = _pair(a.get('pads', 0))
            attrs = {'kernel_shape': _pair(a['kernel_shape']), 'strides': _pair(a.get('strides', 1)), 'pads': p + p}
        elif op.type == 'Gemm': attrs = {'alpha': 1.0, 'beta': 1.0, 'transA': 0, 'transB': 1}
        elif op.type == 'Flatten': attrs = {'axis': 1}
        elif op.type in ('Relu', 'Add', 'GlobalAveragePool'): attrs = {}
        elif op.type == 'Concat': attrs = {'axis': int(a.get('axis', 1))}
        elif op.type == 'Resize': attrs = {'mode': a.get('mode', 'nearest'), 'scale': a.get('scale', 2)}       # topology only: not executable there
        else: raise NotImplementedError(f'to_reference_graph: {op.type}')
        g.create_operation(op_type=op.type, name=op.name, attributes=attrs,
                           inputs=[var(v) for v in op.inputs], outputs=[var(v) for v in op.outputs])
    for name in hgraph.inputs: g.mark_variable_as_graph_input(made[name])
    for name in hgraph.outputs: g.mark_variable_as_graph_output(made[name])
    return g


def quantize_reference_graph(g, device: str, sample: torch.Tensor, bins: int = 2048, method: Optional[str] = 'kl',
                             mutate=None, platform: str = 'TRT_INT8', parameter_pass=None):
    """The reference's own front half of quantize_native_model (api/interface.py:453-543) with the
    TensorRT INT8 quantizer: dispatch, per-op TQCs, QuantizeSimplifyPass, QuantizeFusionPass,
    ParameterQuantizePass.  Activation configs get `method` and the BASELINE's 2048-bin override
    (range.py:152-153); `method=None` keeps the quantizer's own algorithm (TRT_FP8: 'floating').  `platform`: a
    TargetPlatform name; `parameter_pass`: a pass instance to run in place of the reference's ParameterQuantizePass.
    Returns (graph, executor) ready for RuntimeCalibrationPass."""
    import ppq.lib as PFL
    from ppq import TargetPlatform, TorchExecutor
    from ppq.api import dispatch_graph
    from ppq.core import OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE
    from ppq.quantization.optim import ParameterQuantizePass, QuantizeFusionPass, QuantizeSimplifyPass
    target = getattr(TargetPlatform, platform)
    g = dispatch_graph(g, target, 'conservative')
    quantizer = PFL.Quantizer(platform=target, graph=g)
    TorchExecutor(g, device=device).tracing_operation_meta(inputs=sample.to(device))
    for op in list(g.operations.values()):
        if op.platform not in (TargetPlatform.FP32, TargetPlatform.SOI):
            quantizer.quantize_operation(op.name, platform=op.platform)
    for op in g.operations.values():
        if not hasattr(op, 'config'): continue
        for cfg, v in op.config_with_variable:
            if not v.is_parameter and method is not None:
                cfg.observer_algorithm = method
                cfg.detail[OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE] = bins
            if mutate is not None: mutate(cfg, v)
    ex = TorchExecutor(g, device=device)
    PFL.Pipeline([QuantizeSimplifyPass(), QuantizeFusionPass(activation_type=quantizer.activation_fusion_types),
                  ParameterQuantizePass() if parameter_pass is None else parameter_pass]).optimize(graph=g, dataloader=[sample], executor=ex, calib_steps=8,
                                                     collate_fn=None, verbose=False)
    return g, ex


def quantize_reference_topology(g):
    """Dispatch + per-op TQCs WITHOUT tracing or running anything (for graph-only algorithms such as the block
    builder; lets topologies through whose ops the conversion above cannot make executable, e.g. Resize)."""
    import ppq.lib as PFL
    from ppq import TargetPlatform
    from ppq.api import dispatch_graph
    g = dispatch_graph(g, TargetPlatform.TRT_INT8, 'conservative')
    quantizer = PFL.Quantizer(platform=TargetPlatform.TRT_INT8, graph=g)
    for op in list(g.operations.values()):
        if op.platform not in (TargetPlatform.FP32, TargetPlatform.SOI):
            quantizer.quantize_operation(op.name, platform=op.platform)
    return g


def calibrate(g, ex, batches: List[torch.Tensor], method: Optional[str] = 'kl') -> float:
    """The reference's RuntimeCalibrationPass over `batches`; returns the seconds it took."""
    from ppq.quantization.optim import RuntimeCalibrationPass
    p = RuntimeCalibrationPass(method=method)
    t0 = time.perf_counter()
    p.optimize(graph=g, dataloader=batches, executor=ex, calib_steps=max(8, len(batches)), collate_fn=None)
    return time.perf_counter() - t0


def lsq_finetune(g, ex, batches: List[torch.Tensor], steps: int, lr: float, block_size: int = 5, device: str = 'cuda'):
    """The reference's LearnedStepSizePass (optim/training.py:569-863), unmodified, over `batches`.
    Returns [(start op, end op, [member ops], pre_loss, post_loss)] in block order (its finetune() wrapped to record)."""
    from ppq.quantization.optim import LearnedStepSizePass
    p = LearnedStepSizePass(steps=steps, lr=lr, block_size=block_size, collecting_device=device)
    report, inner = [], p.finetune

    def recording(*a, **k):
        pre, post = inner(*a, **k)
        b = k['block']
        report.append((b.sp.name, b.ep.name, [o.name for o in b.rps], float(pre), float(post)))
        return pre, post
    p.finetune = recording
    p.optimize(graph=g, dataloader=batches, executor=ex, collate_fn=None)
    return report


def bias_correction(g, ex, batches: List[torch.Tensor], block_size: int = 1, device: str = 'cuda'):
    """The reference's BiasCorrectionPass (optim/training.py:338-566), unmodified.  Returns
    ([(start op, pre_loss, post_loss, [member ops])], {bias variable name: value after the pass})."""
    from ppq.quantization.optim import BiasCorrectionPass
    p = BiasCorrectionPass(block_size=block_size, steps=len(batches), collecting_device=device)
    report, inner = [], p.correct_bias

    def recording(*a, **k):
        pre, post = inner(*a, **k)
        report.append((k['block'].sp.name, float(pre), float(post), [o.name for o in k['block'].rps]))
        return pre, post
    p.correct_bias = recording
    p.optimize(graph=g, dataloader=batches, executor=ex, collate_fn=None)
    biases = {op.inputs[-1].name: op.inputs[-1].value.detach().clone() for op in g.operations.values()
              if op.type in ('Conv', 'Gemm', 'ConvTranspose') and len(op.inputs) == 3}
    return report, biases


def all_scales(g) -> Dict[str, List[float]]:
    """{"op:variable": scale values} of EVERY activated config (activations and parameters)."""
    from ppq.core import QuantizationStates
    out = {}
    for op in g.operations.values():
        if not hasattr(op, 'config'): continue
        for cfg, v in op.config_with_variable:
            if cfg.state == QuantizationStates.ACTIVATED and cfg.scale is not None:
                out[f'{op.name}:{v.name}'] = [float(s) for s in cfg.scale.flatten().tolist()]
    return out


def activation_scales(g) -> Dict[str, float]:
    """{variable name: rendered per-tensor scale} of every ACTIVATED activation config."""
    from ppq.core import QuantizationStates
    out = {}
    for op in g.operations.values():
        if not hasattr(op, 'config'): continue
        for cfg, v in op.config_with_variable:
            if not v.is_parameter and cfg.state == QuantizationStates.ACTIVATED and cfg.scale is not None:
                out[v.name] = float(cfg.scale.flatten()[0])
    return out


def timed_calibrate(hgraph, batches: List[torch.Tensor], bins: int = 2048):
    """bench.py cpu_baseline, kind "reference": the reference's CPU path (USING_CUDA_KERNEL False, its
    default) calibrating the same topology on the same kind of batches.  Like the GPU leg, only
    RuntimeCalibrationPass.optimize is inside the timer.  The reference asserts calib_steps >= 8
    (optim/calibration.py:136-139) and cycles its dataloader to reach it, so fewer than 8 batches are
    visited more than once; the returned seconds are scaled to ONE visit of `batches`."""
    load()
    from ppq.core import PPQ_CONFIG
    assert PPQ_CONFIG.USING_CUDA_KERNEL is False
    g, ex = quantize_reference_graph(to_reference_graph(hgraph), 'cpu', batches[0], bins)
    steps = max(8, len(batches))
    secs = calibrate(g, ex, batches)
    return secs * len(batches) / steps, activation_scales(g)


class ReplayExecutor:
    """A deterministic stand-in for ``ppq.TorchExecutor`` in calibration comparisons -- TEST INFRASTRUCTURE.

    The vendor convolutions are not bit-reproducible from one forward to the next on large topologies (the algorithm
    picked depends on the workspace the allocator can offer at that moment), so two calibration passes that each run
    their own forwards can see activations that differ in the last bit -- enough to move one histogram count and flip a
    KL arg-min.  This wrapper runs the REFERENCE executor once per batch, records what its hooks are shown (the raw and
    quantised inputs / outputs of every quantable operation, executor/torch.py:523-549), and afterwards REPLAYS exactly
    those tensors into whatever hooks a pass hands it -- the reference's pass with its own observers, the reference's
    pass with this package's observers, this package's pass: all see identical bits.  Valid while no activation config is
    switched on between the recording and a replay (true for every calibration pass: activations stay INITIAL until the
    last render; `reset_activation_configs` restores that state between passes).

    ``target_graph``: replay into hooks built on ANOTHER graph with the same operation names (a ppq_amd.harness graph):
    the recorded configs are translated by (operation, position)."""
    def __init__(self, inner, target_graph=None):
        load()
        self._inner = inner
        self._ref_graph = inner._graph
        self._graph = target_graph if target_graph is not None else inner._graph
        self._translate = target_graph is not None
        self._tape: Dict[int, list] = {}
        self.recorded_forwards = 0
        self.replayed_forwards = 0

    def _record(self, inputs) -> list:
        from ppq.executor.base import QuantOPRuntimeHook
        from ppq.IR import QuantableOperation
        tape = []

        class Spy(QuantOPRuntimeHook):
            def __init__(self, op): self._op = op; super().__init__(op)

            def pre_forward_hook(self, inputs, quant_inputs, quant_configs):
                tape.append((self._op.name, 'pre', list(inputs), list(quant_inputs), list(quant_configs)))
                return quant_inputs

            def post_forward_hook(self, outputs, quant_outputs, quant_configs):
                tape.append((self._op.name, 'post', list(outputs), list(quant_outputs), list(quant_configs)))
                return quant_outputs
        spies = {name: Spy(op) for name, op in self._ref_graph.operations.items() if isinstance(op, QuantableOperation)}
        self._inner.forward(inputs=inputs, hooks=spies)
        self.recorded_forwards += 1
        return tape

    def forward(self, inputs, output_names=None, hooks=None):
        key = (inputs.data_ptr(), tuple(inputs.shape))
        if key not in self._tape: self._tape[key] = self._record(inputs)
        self.replayed_forwards += 1
        for name, kind, raw, quant, cfgs in self._tape[key]:
            hook = hooks.get(name) if hooks else None
            if hook is None: continue
            if self._translate:
                op = self._graph.operations[name]
                cfgs = list(op.config.input_quantization_config if kind == 'pre' else op.config.output_quantization_config)
            if kind == 'pre': hook.pre_forward_hook(inputs=raw, quant_inputs=quant, quant_configs=cfgs)
            else: hook.post_forward_hook(outputs=raw, quant_outputs=quant, quant_configs=cfgs)
        return []

    def for_graph(self, target_graph) -> 'ReplayExecutor':
        """A view on the SAME recording that replays into hooks built on `target_graph` (same operation names)."""
        view = ReplayExecutor(self._inner, target_graph)
        view._tape = self._tape
        return view

    def reset_activation_configs(self):
        """Put every non-parameter config of the (reference) graph back to INITIAL with no scale: the next pass starts
        from the state the recording was made in."""
        from ppq.core import QuantizationStates
        from ppq.IR import QuantableOperation
        for op in self._ref_graph.operations.values():
            if not isinstance(op, QuantableOperation): continue
            for cfg, v in op.config_with_variable:
                if not v.is_parameter and cfg.state == QuantizationStates.ACTIVATED and cfg.dominated_by == cfg:
                    cfg.state = QuantizationStates.INITIAL
                    cfg.scale = None; cfg.offset = None
