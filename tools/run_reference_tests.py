"""Run the reference's own kernel tests, UNCHANGED, against libppq_hip.so.

    python tools/run_reference_tests.py /path/to/ppq-checkout

Needs both an MI355X and an importable PPQ checkout (neither environment of this project has
both: the build container has no GPU, the GPU box has no /root/reference), so this is the recipe a
maintainer runs; the same shapes, distributions and tolerances are replayed by
tests/test_gpu_kernels.py against the oracle.

It (1) stubs `onnx` if missing, (2) makes ComplieHelper.complie install ppq_amd.HIP_EXTENSION
instead of JIT-building ppq/csrc with nvcc, (3) executes tests/test_cuda_kernel.py and
tests/test_rounding.py with runpy -- every `CUDA.*` call in them lands in the HIP kernels.
"""
import importlib.machinery
import os
import runpy
import sys
from unittest.mock import MagicMock


def main(ref: str) -> None:
    os.environ.setdefault('PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION', 'python')
    try:
        import onnx  # noqa: F401
    except ImportError:
        for name in ['onnx', 'onnx.helper', 'onnx.numpy_helper', 'onnx.mapping', 'onnx.onnx_pb', 'onnx.checker',
                     'onnx.external_data_helper', 'onnx.shape_inference', 'onnx.version_converter']:
            m = MagicMock(); m.__spec__ = importlib.machinery.ModuleSpec(name, None); m.__path__ = []
            sys.modules[name] = m
    sys.path.insert(0, ref)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ppq_amd
    from ppq.core import PPQ_CONFIG
    from ppq.core import ffi as ref_ffi

    def complie(self):                      # ppq/core/ffi.py:20-41 without the nvcc JIT build
        self.__CUDA_EXTENTION__ = ppq_amd.HIP_EXTENSION
    ref_ffi.ComplieHelper.complie = complie
    ref_ffi.CUDA_COMPLIER.complie()
    PPQ_CONFIG.USING_CUDA_KERNEL = True
    for test in ('tests/test_cuda_kernel.py', 'tests/test_rounding.py'):
        print(f'== {test}')
        runpy.run_path(os.path.join(ref, test), run_name='__main__')
    print('reference kernel tests passed on libppq_hip.so')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '/root/reference')
