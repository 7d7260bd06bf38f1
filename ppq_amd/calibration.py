"""RuntimeCalibrationPass -- the named hot path -- over the HIP observers.

Mirror of ppq/quantization/optim/calibration.py:19-213 behind the plugin API of
ppq/quantization/optim/base.py:8-28: ``optimize(graph, dataloader, executor, calib_steps,
collate_fn, **kwargs)``, same two-phase structure, same assertions on ``calib_steps``, same
``override`` semantics, same rule for dropping one-phase observers before phase 2.

MI355X-first differences (results unchanged):

* observers accumulate into device buffers through the HIP kernels (ppq_amd/observer.py);
* ``batch_observations`` (default): the statistics of every tensor observed during a forward are
  computed by ONE multi-tensor launch at the end of that forward (observer.ObservationQueue);
* ``use_hip_graph`` (False | True | 'auto'): each phase's forward is captured once into a HIP graph
  and replayed -- 'auto' times one eager step and captures only when it is launch-bound (small batches);
* ``reuse_activations`` (opt-in): phase 2 bins the phase-1 activations kept in HBM instead of running
  the forward again (valid while phase-1 rendering activates no activation config; checked);
* optionally (``async_observe``) the observer kernels run on a side HIP stream (they only read);
* both render steps go through :func:`ppq_amd.observer.render_observers`: one device->host copy
  for all running ranges, one batched KL / MSE search launch per group of histograms;
* data-parallel calibration: with ``torch.distributed`` initialised (one process per GPU, RCCL over
  xGMI) every rank calibrates on its own shard of the batches and the per-phase statistics are
  merged by :func:`ppq_amd.distributed.merge_observers` -- ONE flat all-reduce per reduction kind
  and phase -- before rendering, so every rank renders identical scales.
"""
import time
from math import ceil
from typing import Callable, Dict, Iterable, List

from .core import QuantizationStates, state_value
from .distributed import merge_observers
from .observer import OperationObserver, TorchHistObserver, render_observers


class QuantizationOptimizationPass:
    """ppq/quantization/optim/base.py:8-28."""
    def __init__(self, name: str = 'Default Quanzation Optim') -> None:
        self.name = name

    def apply(self, graph, **kwargs) -> None:
        self.optimize(graph, **kwargs)

    def optimize(self, graph, **kwargs) -> None:
        raise NotImplementedError('Implement this function first.')

    def __str__(self) -> str:
        return 'QuantizationOptimizationPass[' + self.name + ']'


class RuntimeCalibrationPass(QuantizationOptimizationPass):
    # use_hip_graph='auto' captures a phase when enqueueing a step takes more than this fraction of
    # its synchronised wall time and at least this many steps remain to amortise capture + instantiate
    AUTO_GRAPH_ISSUE_FRACTION = 0.7
    AUTO_GRAPH_MIN_STEPS = 12

    def __init__(self, method: str = None, override: bool = False, calib_steps: int = 32,
                 process_group=None, check_steps: bool = True, async_observe: bool = False,
                 use_hip_graph: bool = False, batch_observations: bool = True,
                 reuse_activations: bool = False, reuse_budget_bytes: int = 64 << 30,
                 queue_bytes: int = None) -> None:
        super().__init__(name='PPQ Runtime Calibration Pass')
        self._method = method
        self._observers: Dict[str, OperationObserver] = {}
        self._collate_fn = None
        self._calib_steps = calib_steps
        self._override = override
        self._process_group = process_group
        self._check_steps = check_steps
        self._async_observe = async_observe
        self._use_hip_graph = use_hip_graph
        self._batch_observations = batch_observations
        self._queue = None
        self._reuse_activations = reuse_activations
        self._queue_bytes = queue_bytes
        self._reuse_budget = reuse_budget_bytes
        self._replay: list = []            # per phase-1 batch: [(hist observer, activation tensor)]
        self._replay_bytes = 0
        self.replay_peak_bytes = 0          # most activation bytes ever kept resident for a phase-2 replay
        self.replayed_batches = 0
        self._recording = False
        self.graph_replays = 0
        self.graph_decisions = []      # use_hip_graph='auto': one record per phase
        self.merge_stats = []          # data-parallel runs: one record per phase (collectives, bytes, ms)
        self._side_stream = None

    def _batches(self, dataloader: Iterable) -> list:
        """The sequence of batches the reference loop (calibration.py:108-121) would feed."""
        out = []
        for calib_epoch in range(ceil(self._calib_steps / len(dataloader))):
            for data in dataloader:
                if self._collate_fn is not None:
                    data = self._collate_fn(data)
                out.append(data)
                if len(out) >= self._calib_steps: return out
        return out

    def _graph_replayable(self, batches: list, hooks: Dict[str, object]) -> bool:
        """A calibration forward can be captured into a HIP graph when it is a fixed kernel sequence:
        same-shaped CUDA batches and observers whose whole state lives in device buffers."""
        import torch
        from .observer import ConstantObserver, TorchHistObserver, TorchMinMaxObserver, TorchMSEObserver
        if not self._use_hip_graph or len(batches) < 4: return False
        if not all(isinstance(b, torch.Tensor) and b.is_cuda and b.shape == batches[0].shape
                   and b.dtype == batches[0].dtype for b in batches): return False
        safe = (TorchMinMaxObserver, TorchHistObserver, TorchMSEObserver, ConstantObserver)
        return all(type(ob) in safe for hook in hooks.values() for ob in hook._observer_table.values())

    @ staticmethod
    def _forward_fn(executor):
        """What one calibration forward calls: the executor's public ``forward``.  (The reference's ``TorchExecutor.forward``,
        executor/torch.py:365-410, is its ``forward_with_gradient`` under ``@torch.no_grad()`` -- the same loop over
        ``_executing_order``; only ``tracing_operation_meta`` (:579-580) carries ``@empty_ppq_cache``.  Rounds 4-5 called
        ``forward_with_gradient`` here in the belief that ``forward`` emptied the allocator before every batch; it does not, and
        the batch-1 seam figures (94.7 -> 325 samples/s) come from the HIP-graph replay of that loop, not from a skipped flush.)"""
        return executor.forward

    def _forward(self, executor, data, hooks, output_names):
        import torch
        recording = self._queue is not None and self._recording
        if recording: self._queue.recorder = []
        with torch.no_grad():
            self._forward_fn(executor)(inputs=data, hooks=hooks, output_names=output_names)
        if self._queue is not None: self._queue.flush()      # one multi-tensor launch per statistic kind
        if recording:
            rec, self._queue.recorder = self._queue.recorder, None
            size = sum(v.numel() * 4 for _, v in rec)
            if self._replay_bytes + size <= self._reuse_budget:
                self._replay.append(rec); self._replay_bytes += size
                self.replay_peak_bytes = max(self.replay_peak_bytes, self._replay_bytes)
            else: self._recording = False                     # budget reached: later batches run their forward again

    def _replay_phase2(self, batches: list) -> int:
        """Phase 2 over the activations kept from phase 1 (``reuse_activations``): the second forward of
        the reference recomputes exactly the tensors phase 1 saw -- as long as no activation config was
        activated by the phase-1 render and the executor is deterministic -- so with 288 GB of HBM they
        are simply kept (ResNet-50, 256 samples: 17 GB) and binned from memory.  Returns how many leading
        batches were served from memory; the rest run the normal forward."""
        n = min(len(self._replay), len(batches))
        for i in range(n):
            for ob, value in self._replay[i]: ob.observe(value)
            self._queue.flush()
            self._replay[i] = None                            # release the batch's activations
        self._replay, self._replay_bytes = [], 0
        self.replayed_batches += n
        return n

    def calibrate(self, desc: str, dataloader: Iterable, executor, hooks: Dict[str, object],
                  output_names: List[str] = None):
        """calibration.py:105-121 (progress bar omitted).

        Optional (``use_hip_graph``): when the batches have a fixed shape the forward (dense ops +
        observer kernels, incl. the side-stream fork) is captured ONCE per phase into a HIP graph and
        replayed for the remaining batches, which takes the Python / launch overhead off the critical
        path.  Batch 0 runs eagerly (it allocates the observer buffers and lets MIOpen pick its
        kernels), then the step is captured and the remaining batches replay.  Measured on MI355X
        (ResNet-50, 256 samples): batch 1 x 256 steps 127 -> 326 samples/s, batch 8 x 32 steps
        973 -> 1385, batch 32 x 8 steps: the eager loop is already GPU-bound and capture costs more
        than it saves.  ``use_hip_graph='auto'`` therefore times one eager step and captures only
        when the loop is launch-bound (profiles/HISTORY.md section 6)."""
        import torch
        batches = self._batches(dataloader)
        if self._replay and not self._recording:
            batches = batches[self._replay_phase2(batches):]
            if not batches: return
        if not self._graph_replayable(batches, hooks):
            for data in batches:
                self._forward(executor, data, hooks, output_names)
            return
        self._forward(executor, batches[0], hooks, output_names)
        first = 1
        if self._use_hip_graph == 'auto':
            # Is this loop launch-bound?  Time batch 1: `issue` = host time to enqueue the step,
            # `total` = until the GPU has drained it.  A GPU-bound step leaves the host waiting
            # (issue << total) and a graph cannot help; a launch-bound one (small batches: hundreds
            # of microsecond kernels) keeps the GPU idle between launches (issue ~ total).
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self._forward(executor, batches[1], hooks, output_names)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            first = 2
            launch_bound = (t1 - t0) > self.AUTO_GRAPH_ISSUE_FRACTION * (t2 - t0)
            self.graph_decisions.append({'phase': desc, 'issue_ms': (t1 - t0) * 1e3, 'total_ms': (t2 - t0) * 1e3,
                                         'graph': bool(launch_bound and len(batches) - first >= self.AUTO_GRAPH_MIN_STEPS)})
            if not self.graph_decisions[-1]['graph']:
                for data in batches[first:]:
                    self._forward(executor, data, hooks, output_names)
                return
        self._recording = False          # tensors produced inside a captured graph are overwritten by every replay
        static_in = torch.empty_like(batches[0])
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._forward(executor, static_in, hooks, output_names)
            if self._side_stream is not None:
                torch.cuda.current_stream().wait_stream(self._side_stream)
        for data in batches[first:]:
            static_in.copy_(data, non_blocking=True)
            graph.replay()
            self.graph_replays += 1

    def _all_tensor_observers(self):
        return [ob for op_ob in self._observers.values() for ob in op_ob.observers()]

    def _render(self):
        observers = self._all_tensor_observers()
        if self._queue is not None: self._queue.flush()
        if self._side_stream is not None:          # join the observer stream before reading statistics
            import torch
            torch.cuda.current_stream().wait_stream(self._side_stream)
        if merge_observers(observers, group=self._process_group):
            from . import distributed
            self.merge_stats.append(dict(distributed.last_merge_stats))
        render_observers(observers)

    def optimize(self, graph, dataloader: Iterable, executor, calib_steps: int = 32,
                 collate_fn: Callable = None, **kwargs) -> None:
        if collate_fn is not None: self._collate_fn = collate_fn
        if calib_steps is not None: self._calib_steps = calib_steps
        if self._check_steps:
            assert self._calib_steps >= 8, (
                'Insufficient Calibration Detected, to get a better quantization performance, '
                'more calibration steps is required, we strongly recommend you to prepare more calibration data '
                'and more calibration steps is preferred here. (at least 8)')
            assert self._calib_steps <= 512, (
                'Calibration steps is too large, ppq can quantize your network within 8-512 '
                'calibration steps. More calibration steps will greatly delay ppq\'s calibration procedure. '
                'Reset your calib_steps parameter please.')

        # override existing quantization configurations (calibration.py:147-155)
        if self._override:
            for operation in graph.operations.values():
                if not hasattr(operation, 'config'): continue
                for config, var in operation.config_with_variable:
                    if (not var.is_parameter and state_value(config.state) == QuantizationStates.ACTIVATED.value
                            and config.dominated_by == config):
                        config.state = type(config.state).INITIAL

        # build observer and hook for each quantable operation (calibration.py:157-172)
        self._observers = {}
        hooks = {}
        import torch
        if self._async_observe and torch.cuda.is_available():
            self._side_stream = torch.cuda.Stream()
        for op_name, operation in graph.operations.items():
            if not hasattr(operation, 'config'): continue
            for config, var in operation.config_with_variable:
                if not var.is_parameter and self._method is not None:
                    config.observer_algorithm = self._method
            observer = OperationObserver(operation=executor._graph.operations[op_name], monitor_parameter=False)
            observer.hook.stream = self._side_stream
            self._observers[op_name] = observer
            hooks[op_name] = observer.hook

        # queue the per-tensor statistics kernels of a forward into one multi-tensor launch; side-stream
        # observation keeps its per-tensor launches (they are ordered against each producer)
        self._queue = None
        if self._batch_observations and self._side_stream is None:
            from .observer import ObservationQueue
            self._queue = ObservationQueue() if self._queue_bytes is None else ObservationQueue(self._queue_bytes)
        for ob in self._all_tensor_observers(): ob.queue = self._queue

        self._replay, self._replay_bytes = [], 0
        self._recording = bool(self._reuse_activations and self._queue is not None)
        self.calibrate(desc='Calibration Progress(Phase 1)', dataloader=dataloader, executor=executor,
                       hooks=hooks, output_names=None)
        self._recording = False
        self._render()
        if self._replay:
            # the kept activations are only what a second forward would produce if phase-1 rendering did
            # not switch any ACTIVATION config on (a mixed minmax / kl graph does): otherwise drop them
            for observer in self._observers.values():
                for cfg, ob in observer.hook._observer_table.items():
                    if (not getattr(ob._watch_on, 'is_parameter', False)
                            and QuantizationStates.is_activated(cfg.state)):
                        self._replay, self._replay_bytes = [], 0
                        break
                if not self._replay: break

        # remove one-phase observers (calibration.py:192-201)
        pop_list = []
        for op_name, observer in self._observers.items():
            if all([not isinstance(var_observer, TorchHistObserver)            # kl, mse, kl_channel: two-phase
                    for var_observer in observer.observers()]):
                pop_list.append(op_name)
        for op_name in pop_list:
            self._observers.pop(op_name)
            hooks.pop(op_name)

        if len(hooks) > 0:
            self.calibrate(desc='Calibration Progress(Phase 2)', dataloader=dataloader, executor=executor,
                           hooks=hooks, output_names=None)
            self._render()


class PPLDSPTIReCalibrationPass(RuntimeCalibrationPass):
    """optim/calibration.py:216-322: the extra calibration the PPL DSP TI platforms ask for -- a PER-CHANNEL running range of
    every computing operation's output (of the Relu / Clip behind it when that is its only consumer), plus a per-tensor range of
    the graph's input where a computing operation reads it; nothing is rendered, the ranges are recorded in the consumer config's
    ``detail['range_min' / 'range_max']``.  The observers are this package's device-resident min/max observers (per channel:
    ``ppqhip_minmax_c``, no transpose-and-flatten copy per batch as in observer/range.py:99-107), read back with one copy each at
    the end.  The per-tensor entries are 1-tuples, as the reference writes them (calibration.py:319-320 end in a comma)."""
    def __init__(self, method: str = None, override: bool = False) -> None:
        super().__init__(method, override)
        self.name = 'PPQ ReCalibration For Computing Op Pass'

    def optimize(self, graph, dataloader: Iterable, executor, calib_steps: int, collate_fn: Callable = None, **kwargs) -> None:
        from .blocks import COMPUTING_OP, downstream_operations
        from .core import QuantizationPolicy, QuantizationProperty as P, TensorQuantizationConfig
        from .observer import CalibrationHook, TensorObserverFactroy, TorchMinMaxObserver
        self._collate_fn, self._calib_steps = collate_fn, calib_steps
        assert calib_steps >= 8, ('Insufficient Calibration Detected, to better quantize your network, more calibration steps is '
                                  'demonded, we strongly recommend you to prepare more calibration data and more calibration '
                                  'steps is preferred here. (at least 8)')
        assert calib_steps <= 512, ('Calibration steps is too large, ppq is capable for quantizing your network within 32-128 '
                                    'calibration steps. More calibraiton steps will greatly delay ppq\'s calibration procedure. '
                                    'Reset your calib_steps parameter please.')

        def probe(like, per_channel: bool, consumer):
            bits = P.SYMMETRICAL.value + P.LINEAR.value + (P.PER_CHANNEL.value if per_channel else P.PER_TENSOR.value)
            return TensorQuantizationConfig(policy=QuantizationPolicy(bits), rounding=like.rounding, num_of_bits=like.num_of_bits,
                                            quant_min=like.quant_min, quant_max=like.quant_max, scale=None, offset=None,
                                            observer_algorithm='Minmax', state=QuantizationStates.INITIAL,
                                            channel_axis=1 if per_channel else None, detail={'consumer': consumer})
        hooks = {}
        for operation in graph.topological_sort():
            if not hasattr(operation, 'config') or operation.type not in COMPUTING_OP: continue
            output_cfg = operation.config.output_quantization_config[0]
            master_cfg, master_operation, master_var = output_cfg, operation, operation.outputs[0]
            table = {}
            if operation.inputs[0].name in graph.inputs:            # is all input data greater than 0? a basic range suffices
                input_cfg = operation.config.input_quantization_config[0]
                table[input_cfg] = TensorObserverFactroy.build_observer(operation.inputs[0], probe(input_cfg, False, input_cfg))
            followers = downstream_operations(operation)
            if len(followers) == 1 and followers[0].type in {'Relu', 'Clip'} and hasattr(followers[0], 'config'):
                if table:
                    hooks[operation.name] = CalibrationHook(operation, table)
                    table = {}
                master_operation = followers[0]
                master_cfg = master_operation.config.output_quantization_config[0]
                master_var = master_operation.outputs[0]
            table[master_cfg] = TensorObserverFactroy.build_observer(master_var, probe(master_cfg, True, output_cfg))
            assert master_operation.name not in hooks, 'register an operation in calibration hooks twice'
            hooks[master_operation.name] = CalibrationHook(master_operation, table)

        self._queue = None
        self.calibrate(desc='ReCalibration For Computing Ops', dataloader=dataloader, executor=executor, hooks=hooks)

        for hook in hooks.values():
            for observer in hook._observer_table.values():
                cfg = observer._quant_cfg.detail['consumer']
                assert isinstance(observer, TorchMinMaxObserver)
                r = observer._range_on_host()
                if observer._quant_cfg.policy.has_property(P.PER_CHANNEL):
                    cfg.detail.update({'range_min': r[0].copy(), 'range_max': r[1].copy()})
                else:
                    cfg.detail.update({'range_min': (float(r[0]),), 'range_max': (float(r[1]),)})


class IsotoneCalibrationPass(RuntimeCalibrationPass):
    """optim/calibration.py:325-422.  Marks classification outputs for the order-preserving 'isotone' observer -- by default the
    output of every Softmax that owns its config, otherwise the variables named in ``variables`` with ``axis`` as the class
    axis -- and then CALIBRATES them: like the reference's, this pass ends in ``RuntimeCalibrationPass.optimize`` (method None,
    so the 'Isotone' marks are kept, every other INITIAL config is observed with its own algorithm)."""
    def __init__(self, variables: List[str] = None, axis: int = -1, verbose: bool = True, calib_steps: int = 32) -> None:
        super().__init__(calib_steps=calib_steps)
        self.name = 'Isotone Calibration Pass'
        self.variables = variables
        self.axis = axis
        self.verbose = verbose

    @ staticmethod
    def _mark(cfg, axis: int) -> None:
        from .core import OBSERVER_ISOTONE_OBSERVER_AXIS
        cfg.state = type(cfg.state).INITIAL                               # the graph may carry the reference's own enum
        cfg.observer_algorithm = 'Isotone'
        cfg.detail[OBSERVER_ISOTONE_OBSERVER_AXIS] = axis

    def optimize(self, graph, **kwargs) -> None:
        if self.variables is None:
            for op in graph.operations.values():
                if op.type != 'Softmax' or not hasattr(op, 'config'): continue
                cfg = op.config.output_quantization_config[0]
                if cfg.dominated_by != cfg: continue                          # a dominated config follows its master
                axis = op.attributes.get('axis', -1)
                self._mark(cfg, axis)
                if self.verbose: print(f'Calibration Method of Op {op.name} has been changed to Isotone[axis={axis}].')
        else:
            if not isinstance(self.variables, list):
                raise TypeError('Isotone Calibration Pass needs a list of variable name as its input.')
            for name in self.variables:                                        # checked one by one, as the reference does:
                if not isinstance(name, str):                                  # names before a bad entry ARE marked
                    raise TypeError('Isotone Calibration Pass needs a list of variable name as its input.')
                if name not in graph.variables: raise ValueError(f'Variable {name} not in current graph.')
                var = graph.variables[name]
                op = var.source_op
                if op is None or not hasattr(op, 'config'): continue           # not a QuantableVariable
                self._mark(op.config.output_quantization_config[op.outputs.index(var)], self.axis)
                if self.verbose: print(f'Calibration Method of Variable {var.name} has been changed to Isotone[axis={self.axis}].')
        super().optimize(graph, **kwargs)

