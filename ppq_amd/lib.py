"""The slice of ``ppq.lib`` (PFL) that sits on this path -- ppq/lib/quant.py:38-166, ppq/lib/extension.py:76-93.

``Observer`` / ``QuantFunction`` / ``Pipeline`` / the two quant stubs / the config constructors, with the reference's names and
call protocol, so that code written as ``import ppq.lib as PFL`` against these entry points reads the same with
``import ppq_amd.lib as PFL``.  Quantizers, dispatchers, parsers and exporters are PPQ's control plane and stay there.
"""
from typing import Callable, List

import torch

from .calibration import QuantizationOptimizationPass
from .core import FloatingQuantizationConfig, LinearQuantizationConfig  # noqa: F401  (re-exported, lib/quant.py:106-166)
from .observer import BaseTensorObserver, TensorObserverFactroy, register_calibration_observer  # noqa: F401
from .qfunction import PPQuantFunction as QuantFunction


class QuantizationOptimizationPipeline:
    """ppq/quantization/optim/base.py:31-98: an ordered collection of passes; ``optimize`` hands every pass the graph and the
    same keyword arguments (``dataloader``, ``executor``, ``calib_steps``, ``collate_fn`` ...)."""
    def __init__(self, passes: List[QuantizationOptimizationPass]):
        self._pipeline: List[QuantizationOptimizationPass] = []
        for optim in (passes or []): self.append_optimization_to_pipeline(optim_pass=optim)

    def __len__(self) -> int: return len(self._pipeline)
    def __iter__(self): return iter(self._pipeline)

    def __contains__(self, item) -> bool:
        assert isinstance(item, QuantizationOptimizationPass), (
            'Quantization Optimization Pipeline object only suppose to contain optimization passes, '
            f'while you require to check a/an {type(item)} whether in the optimization list')
        return item in self._pipeline

    def optimize(self, graph, verbose: bool = True, **kwargs) -> None:
        import time
        width = max([len(p.name) for p in self._pipeline], default=0)
        for optim_pass in self._pipeline:
            if not isinstance(optim_pass, QuantizationOptimizationPass):
                raise TypeError('Quantization Optimization Pipeline object only suppose to contain optimization passes only, '
                                f'while {str(optim_pass)}({type(optim_pass)}) was found.')
            if verbose:
                print(f'[{time.strftime("%H:%M:%S", time.localtime())}] {optim_pass.name} Running ... '
                      + ' ' * (width - len(optim_pass.name)), end='')
            optim_pass.optimize(graph=graph, **kwargs)
            if verbose: print('Finished.')

    def append_optimization_to_pipeline(self, optim_pass: QuantizationOptimizationPass, at_front: bool = False):
        assert isinstance(optim_pass, QuantizationOptimizationPass), (
            'Quantization Optimization Pipeline object only suppose to contain optimization passes, '
            f'while we got a/an {type(optim_pass)} in the optimization list')
        if at_front: self._pipeline.insert(0, optim_pass)
        else: self._pipeline.append(optim_pass)
        return self

    def report(self) -> str:
        return ''.join(str(p) + '\n' for p in self._pipeline)


def Pipeline(optims: List[QuantizationOptimizationPass]) -> QuantizationOptimizationPipeline:
    """lib/quant.py:38-44."""
    return QuantizationOptimizationPipeline(optims)


def Observer(quant_config, variable=None) -> BaseTensorObserver:
    """lib/quant.py:47-55: the calibration observer ``quant_config.observer_algorithm`` names."""
    return TensorObserverFactroy.build_observer(variable=variable, config=quant_config)


class TensorQuant(torch.nn.Module):
    """lib/quant.py:58-93: a quant stub -- observe batches, render the config, then fake-quantise (through a delegator when one
    is set).  (The reference never calls ``Module.__init__``, so its stub can only be used through ``.forward``; this one is a
    proper module and can be called.)"""
    def __init__(self, quant_config) -> None:
        super().__init__()
        self._quant_config = quant_config
        self._delegator = None
        self._batch_observed = 0
        self._observer = Observer(quant_config=quant_config)

    @ property
    def delegator(self) -> Callable: return self._delegator

    @ delegator.setter
    def delegator(self, func: Callable): self._delegator = func

    def forward(self, value: torch.Tensor) -> torch.Tensor:
        if self._delegator is not None: return self._delegator(value, self._quant_config)
        return QuantFunction(tensor=value, config=self._quant_config)

    def observe(self, value: torch.Tensor):
        self._batch_observed += 1
        self._observer.observe(value)

    def render(self):
        if self._batch_observed == 0:
            raise PermissionError('You have not provide any data to this QuantStub, PPQ can not render its quant config yet.')
        self._observer.render_quantization_config()


class ParameterQuant(TensorQuant):
    """lib/quant.py:96-103: a stub whose config is rendered from the parameter itself at construction."""
    def __init__(self, quant_config, parameter: torch.Tensor) -> None:
        if not isinstance(parameter, torch.Tensor):
            raise TypeError(f'Expect a torch.Tensor here. However {type(parameter)} was given.')
        super().__init__(quant_config)
        self.observe(parameter)
        self.render()


__all__ = ['Observer', 'Pipeline', 'QuantFunction', 'TensorQuant', 'ParameterQuant', 'LinearQuantizationConfig',
           'FloatingQuantizationConfig', 'QuantizationOptimizationPipeline', 'register_calibration_observer']
