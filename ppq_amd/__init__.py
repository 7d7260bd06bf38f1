"""ppq_amd -- MI355X-native (gfx950) implementation of PPQ's quantization-simulation hot path:
fake-quant kernels, calibration statistics (min/max, histograms, quantiles) and the KL / MSE /
percentile clipping searches of ``RuntimeCalibrationPass``, behind PPQ's own ``ppq.core.ffi.CUDA``
operator surface.  Importing this package loads ``libppq_hip.so`` and fails loudly if it is absent.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library has not been built)
from .core import (PPQ_CONFIG, FloatingQuantizationConfig, LinearQuantizationConfig, QuantizationPolicy,
                   QuantizationProperty, QuantizationStates, RoundingPolicy, TensorQuantizationConfig)
from .ffi import (CUDA, CUDA_COMPLIER, ENABLE_CUDA_KERNEL, HIP_EXTENSION, install_into_ppq, install_plugins_into_ppq,
                  uninstall_from_ppq)

__all__ = ['CUDA', 'CUDA_COMPLIER', 'ENABLE_CUDA_KERNEL', 'HIP_EXTENSION', 'install_into_ppq', 'install_plugins_into_ppq', 'uninstall_from_ppq', 'PPQ_CONFIG', 'RoundingPolicy',
           'QuantizationProperty', 'QuantizationPolicy', 'QuantizationStates', 'TensorQuantizationConfig',
           'LinearQuantizationConfig', 'FloatingQuantizationConfig']
