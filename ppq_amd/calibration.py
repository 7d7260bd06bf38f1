"""RuntimeCalibrationPass -- the named hot path -- over the HIP observers.

Mirror of ppq/quantization/optim/calibration.py:19-213 behind the plugin API of
ppq/quantization/optim/base.py:8-28: ``optimize(graph, dataloader, executor, calib_steps,
collate_fn, **kwargs)``, same two-phase structure, same assertions on ``calib_steps``, same
``override`` semantics, same rule for dropping one-phase observers before phase 2.

MI355X-first differences (results unchanged):

* observers accumulate into device buffers through the HIP kernels (ppq_amd/observer.py);
* both render steps go through :func:`ppq_amd.observer.render_observers`: one device->host copy
  for all running ranges, one batched KL / MSE search launch per group of histograms;
* data-parallel calibration: with ``torch.distributed`` initialised (one process per GPU, RCCL over
  xGMI) every rank calibrates on its own shard of the batches and the per-phase statistics are
  merged by :func:`ppq_amd.distributed.merge_observers` -- ONE flat all-reduce per reduction kind
  and phase -- before rendering, so every rank renders identical scales.
"""
from math import ceil
from typing import Callable, Dict, Iterable, List

from .core import QuantizationStates, state_value
from .distributed import merge_observers
from .observer import OperationObserver, TorchHistObserver, TorchMSEObserver, render_observers


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
    def __init__(self, method: str = None, override: bool = False, calib_steps: int = 32,
                 process_group=None, check_steps: bool = True) -> None:
        super().__init__(name='PPQ Runtime Calibration Pass')
        self._method = method
        self._observers: Dict[str, OperationObserver] = {}
        self._collate_fn = None
        self._calib_steps = calib_steps
        self._override = override
        self._process_group = process_group
        self._check_steps = check_steps

    def calibrate(self, desc: str, dataloader: Iterable, executor, hooks: Dict[str, object],
                  output_names: List[str] = None):
        """calibration.py:105-121 (progress bar omitted)."""
        calib_step = 0
        for calib_epoch in range(ceil(self._calib_steps / len(dataloader))):
            for data in dataloader:
                if self._collate_fn is not None:
                    data = self._collate_fn(data)
                executor.forward(inputs=data, hooks=hooks, output_names=output_names)
                calib_step += 1
                if calib_step >= self._calib_steps: break

    def _all_tensor_observers(self):
        return [ob for op_ob in self._observers.values() for ob in op_ob.observers()]

    def _render(self):
        observers = self._all_tensor_observers()
        merge_observers(observers, group=self._process_group)
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
        for op_name, operation in graph.operations.items():
            if not hasattr(operation, 'config'): continue
            for config, var in operation.config_with_variable:
                if not var.is_parameter and self._method is not None:
                    config.observer_algorithm = self._method
            observer = OperationObserver(operation=executor._graph.operations[op_name], monitor_parameter=False)
            self._observers[op_name] = observer
            hooks[op_name] = observer.hook

        self.calibrate(desc='Calibration Progress(Phase 1)', dataloader=dataloader, executor=executor,
                       hooks=hooks, output_names=None)
        self._render()

        # remove one-phase observers (calibration.py:192-201)
        pop_list = []
        for op_name, observer in self._observers.items():
            if all([type(var_observer) not in {TorchHistObserver, TorchMSEObserver}
                    for var_observer in observer.observers()]):
                pop_list.append(op_name)
        for op_name in pop_list:
            self._observers.pop(op_name)
            hooks.pop(op_name)

        if len(hooks) > 0:
            self.calibrate(desc='Calibration Progress(Phase 2)', dataloader=dataloader, executor=executor,
                           hooks=hooks, output_names=None)
            self._render()
