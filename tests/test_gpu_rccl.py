"""RCCL on the one GPU a test box has: a world-size-1 `nccl` process group -- `nccl` IS RCCL on ROCm -- carries
exactly the buffers `merge_observers` and the data-parallel LSQ step put on the wire, with the (dtype, reduction) pairs they
use.  A 1-rank all-reduce is an identity on the data; what this proves is that librccl loads next to libppq_hip.so, that
it accepts MIN on float32, SUM on int32 / float32 / float64 and MAX on int64, on buffers produced by the HIP observers, on
torch's current stream.  Scaling itself is only measurable on the driver's 8-GPU node.  Runs in a child process so the
process group does not leak into the suite."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', str(29500 + os.getpid() %% 2000))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
assert dist.get_backend() == 'nccl'
import ppq_amd
from ppq_amd import harness, distributed
from ppq_amd.calibration import RuntimeCalibrationPass
from ppq_amd.observer import TensorObserverFactroy
from ppq_amd import LinearQuantizationConfig, FloatingQuantizationConfig

# 1. the real observers' buffers: per-tensor + per-channel ranges (MIN), histograms (SUM int32), percentile sums (SUM f32),
#    FP8 'floating' squared errors (SUM f64); the layout probe is an int64 MAX
g = torch.Generator().manual_seed(0)
xs = [torch.randn(4, 16, 14, 14, generator=g).cuda() for _ in range(3)]
obs = []
for alg, kw in (('minmax', {}), ('minmax', {'channel_axis': 1}), ('kl', {}), ('mse', {}), ('percentile', {})):
    cfg = LinearQuantizationConfig(symmetrical=True, quant_min=-128, quant_max=127, num_of_bits=8, calibration=alg, **kw)
    cfg.detail['OBSERVER_KL_HIST_BINS_MANUL_OVERRIDE'] = 2048
    obs.append(TensorObserverFactroy.build_observer(type('V', (), {'name': alg, 'is_parameter': False})(), cfg))
fcfg = FloatingQuantizationConfig(calibration='floating')
obs.append(TensorObserverFactroy.build_observer(type('V', (), {'name': 'fp8', 'is_parameter': False})(), fcfg))
for x in xs:
    for ob in obs: ob.observe(x)
before = [[b.clone() for b, _ in ob.reducible() if b is not None] for ob in obs]
n1 = distributed.merge_observers(obs, even_if_single_rank=True)
stats1 = dict(distributed.last_merge_stats)
after = [[b for b, _ in ob.reducible() if b is not None] for ob in obs]
for bs, as_ in zip(before, after):
    for b, a in zip(bs, as_): assert torch.equal(b, a), 'a 1-rank all-reduce must be the identity'
from ppq_amd.observer import render_observers
render_observers(obs)                       # phase 1 render: hist observers now know their range
for x in xs:
    for ob in obs: ob.observe(x)            # phase 2: histograms
n2 = distributed.merge_observers(obs, even_if_single_rank=True)
stats2 = dict(distributed.last_merge_stats)
render_observers(obs)
kinds = {k for s in (stats1, stats2) for k in s if k.endswith('_bytes')}
assert {'min_f32_bytes', 'sum_int32_bytes', 'sum_float32_bytes', 'sum_float64_bytes'} <= kinds, kinds
assert stats1['backend'] == 'nccl' and stats1['world_size'] == 1
for ob in obs[:5]: assert float(ob._quant_cfg.scale.flatten()[0]) > 0

# 2. a whole calibration pass with the group initialised: the pass's own merge call sites (world size 1: they return early)
graph = harness.small_cnn_graph(seed=0)
harness.quantize_graph(graph, 'kl', hist_bins=2048)
ex = harness.TorchExecutor(graph, 'cuda')
harness.ParameterQuantizePass().optimize(graph)
batches = [torch.rand(4, 3, 32, 32, generator=g).cuda() for _ in range(8)]
RuntimeCalibrationPass(method='kl').optimize(graph, dataloader=batches, executor=ex, calib_steps=8)

# 3. the data-parallel LSQ gradient exchange: ONE flat float32 SUM all-reduce (ppq_amd/lsq.py::_average)
flat = torch.randn(1 << 20, device='cuda'); ref = flat.clone()
dist.all_reduce(flat, op=dist.ReduceOp.SUM); assert torch.equal(flat, ref)
dist.barrier(device_ids=[0])
torch.cuda.synchronize()
print('RCCL_OK', n1, n2, sorted(kinds))
dist.destroy_process_group()
''' % ROOT


def test_rccl_world_size_one_carries_the_merge_buffers():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', CHILD], capture_output=True, text=True, timeout=900, env=env)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, tail
    maps = subprocess.run(['bash', '-c', 'python - <<"PY"\nimport torch, ctypes.util\nprint(torch.cuda.nccl.version())\nPY'],
                          capture_output=True, text=True, timeout=300)
    print(r.stdout.strip().splitlines()[-1], '| rccl version', maps.stdout.strip())
