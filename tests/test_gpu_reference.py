"""The UNMODIFIED reference running on libppq_hip.so ("drops into ppq.executor
unchanged").  Needs a GPU and an importable reference -- on the GPU box the staged copy of
tools/stage_reference.py (oracle/_ref/ppq_stage, git-ignored); skipped when neither is there."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import reference_import as RI  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(RI.find_reference() is None, reason='no importable reference (stage one)')]
DEV = 'cuda'


def test_reference_kernel_tests_run_unchanged_on_hip():
    """tests/test_cuda_kernel.py and tests/test_rounding.py of the reference, executed with runpy in a
    fresh process; their own asserts decide (bit-exact LT / LC over 12 shapes each, grad_x exact,
    Histogram_T within 100 counts/bin of torch.histc)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_reference_tests.py')], capture_output=True,
                       text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert 'reference kernel tests passed on libppq_hip.so' in r.stdout, tail


@pytest.mark.parametrize('topology,batch,size', [('small_cnn', 4, 32), ('resnet50', 4, 224)])
def test_reference_executor_and_calibration_pass_on_hip(topology, batch, size):
    """The reference's OWN BaseGraph + TensorRT quantizer + TorchExecutor + RuntimeCalibrationPass('kl') on
    the GPU after install_into_ppq() -- vs this package's harness + pass on the same weights and batches:
    the same set of activation configs is calibrated and every rendered scale agrees to 1e-6."""
    import ppq_amd
    from ppq_amd import harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    RI.load()
    ppq_amd.install_into_ppq()
    from ppq.core import PPQ_CONFIG
    from ppq.core.ffi import CUDA_COMPLIER
    assert PPQ_CONFIG.USING_CUDA_KERNEL and CUDA_COMPLIER.CUDA_EXTENSION is ppq_amd.HIP_EXTENSION
    build = harness.small_cnn_graph if topology == 'small_cnn' else harness.resnet50_graph
    g = torch.Generator().manual_seed(7)
    batches = [torch.rand(batch, 3, size, size, generator=g).to(DEV) for _ in range(8)]
    # the reference, unmodified, on our kernels
    counted = {}
    ext = ppq_amd.HIP_EXTENSION
    for name in ('QuantizeTensor_LC', 'QuantizeTensor_LT', 'Histogram_T'):
        def make(fn, name=name):
            def w(*a, **k):
                counted[name] = counted.get(name, 0) + 1
                return fn(*a, **k)
            return w
        setattr(ext, name, make(getattr(type(ext), name)))
    try:
        rg, rex = RI.quantize_reference_graph(RI.to_reference_graph(build(seed=0)), DEV, batches[0], bins=2048)
        RI.calibrate(rg, rex, batches)
        ref_scales = RI.activation_scales(rg)
    finally:
        for name in ('QuantizeTensor_LC', 'QuantizeTensor_LT', 'Histogram_T'): delattr(ext, name)
    assert counted.get('QuantizeTensor_LC', 0) > 0 and counted.get('Histogram_T', 0) > 0, counted
    # this package's harness + pass
    hg = build(seed=0)
    harness.quantize_graph(hg, 'kl', hist_bins=2048)
    hex_ = harness.TorchExecutor(hg, DEV)
    harness.ParameterQuantizePass().optimize(hg)
    RuntimeCalibrationPass(method='kl').optimize(hg, dataloader=batches, executor=hex_, calib_steps=8)
    ours = {v.name: float(c.scale.flatten()[0]) for op in hg.operations.values() if hasattr(op, 'config')
            for c, v in op.config_with_variable
            if not v.is_parameter and int(getattr(c.state, 'value', c.state)) == 4 and c.scale is not None}
    assert set(ours) == set(ref_scales), (sorted(set(ours) ^ set(ref_scales)))
    worst = max(abs(ours[k] - ref_scales[k]) / ref_scales[k] for k in ours)
    assert worst <= 1e-6, (worst, {k: (ours[k], ref_scales[k]) for k in ours if abs(ours[k] - ref_scales[k]) > 1e-6 * ref_scales[k]})
