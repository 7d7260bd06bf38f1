"""Stage an importable copy of the reference for ONE GPU run.

    python tools/stage_reference.py            # /root/reference -> oracle/_ref/ppq_stage  (git-ignored)
    python tools/stage_reference.py --clean    # remove it again

The GPU box has no /root/reference; `gpurun` ships the working tree (minus .git and .gpurunignore'd
paths), and `oracle/_ref/` is git-ignored but travels.  The staged copy is what lets the UNMODIFIED
reference -- its tests/test_cuda_kernel.py, tests/test_rounding.py and its TorchExecutor +
RuntimeCalibrationPass -- execute on libppq_hip.so on a real MI355X (tools/run_reference_tests.py,
tests/test_gpu_reference.py) and what bench.py times as `cpu_baseline.kind == "reference"`.
It is never committed: the repository contains no reference source.
Only what `import ppq` and the two kernel tests need is copied (no samples, docs, models)."""
import os
import shutil
import sys

SRC = '/root/reference'
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'ppq_stage')


def main():
    if '--clean' in sys.argv:
        shutil.rmtree(DST, ignore_errors=True); print('removed', DST); return
    if not os.path.isdir(os.path.join(SRC, 'ppq')): raise SystemExit(f'{SRC} is not here: nothing to stage')
    shutil.rmtree(DST, ignore_errors=True)
    ignore = shutil.ignore_patterns('__pycache__', '*.pyc', 'samples', '*.cu', '*.cuh', '*.cc', '*.h', '*.md', '*.onnx')
    shutil.copytree(os.path.join(SRC, 'ppq'), os.path.join(DST, 'ppq'), ignore=ignore)
    os.makedirs(os.path.join(DST, 'tests'))
    for f in ('test_cuda_kernel.py', 'test_rounding.py'):
        shutil.copy(os.path.join(SRC, 'tests', f), os.path.join(DST, 'tests', f))
    n = sum(len(fs) for _, _, fs in os.walk(DST))
    size = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(DST) for f in fs)
    print(f'staged {n} files, {size / 1e6:.1f} MB -> {DST}')


if __name__ == '__main__':
    main()
