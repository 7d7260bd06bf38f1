"""Run the reference's own kernel tests, UNCHANGED, against libppq_hip.so.

    python tools/run_reference_tests.py [/path/to/ppq-checkout]

Needs an MI355X and an importable reference: on the GPU box that is the staged copy
(`python tools/stage_reference.py` here, then `gpurun`); default = oracle.reference_import.find_reference().
It (1) imports the reference with the shims of oracle/reference_import.py (stub `onnx` if missing, ...),
(2) makes ComplieHelper.complie install ppq_amd.HIP_EXTENSION instead of JIT-building ppq/csrc with
nvcc -- the one line INTEGRATION.md asks a maintainer to change --, (3) executes the reference's
tests/test_cuda_kernel.py and tests/test_rounding.py with runpy: every `CUDA.*` call in them lands in
the HIP kernels, and their own asserts (bit-exact LT / LC, grad checks, <100 counts/bin) decide."""
import collections
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(ref=None) -> None:
    from oracle import reference_import as RI
    ref = RI.load(ref)
    import ppq_amd
    from ppq.core import PPQ_CONFIG
    from ppq.core import ffi as ref_ffi

    calls = collections.Counter()

    class Counting:                      # HIP_EXTENSION with a call counter (proof that the kernels ran)
        def __getattr__(self, name):
            fn = getattr(ppq_amd.HIP_EXTENSION, name)

            def wrapped(*a, **k):
                calls[name] += 1
                return fn(*a, **k)
            return wrapped

    def complie(self):                      # ppq/core/ffi.py:20-41 without the nvcc JIT build
        self.__CUDA_EXTENTION__ = Counting()
    ref_ffi.ComplieHelper.complie = complie
    ref_ffi.CUDA_COMPLIER.complie()
    PPQ_CONFIG.USING_CUDA_KERNEL = True
    print(f'reference: {ref}; library: {ppq_amd._lib.LIB_PATH}')
    for test in ('tests/test_cuda_kernel.py', 'tests/test_rounding.py'):
        print(f'== {test}', flush=True)
        runpy.run_path(os.path.join(ref, test), run_name='__main__')
    print('calls into libppq_hip.so:', dict(calls))
    assert calls['QuantizeTensor_LT'] > 0 and calls['QuantizeTensor_LC'] > 0 and calls['Histogram_T'] > 0
    print('reference kernel tests passed on libppq_hip.so')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else None)
