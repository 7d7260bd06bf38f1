"""Developer aid: where the host time of the 'kernels' seam goes -- the UNMODIFIED reference (executor, pass, observers) on
libppq_hip.so -- by cProfile (cumulative, top entries), ResNet-50 KL 2048 bins, batch 32 x 8."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reference_import as RI  # noqa: E402

import ppq_amd  # noqa: E402
from ppq_amd import harness  # noqa: E402
RI.load()
ppq_amd.install_into_ppq()
from ppq.quantization.optim import RuntimeCalibrationPass as RefPass  # noqa: E402
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1234)
batches = [torch.rand(32, 3, 224, 224, device=dev, generator=g) for _ in range(8)]


def one():
    graph, ex = RI.quantize_reference_graph(RI.to_reference_graph(harness.resnet50_graph(seed=0)), dev, batches[0], bins=2048, method='kl')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    RefPass(method='kl').optimize(graph=graph, dataloader=batches, executor=ex, calib_steps=8, collate_fn=None)
    torch.cuda.synchronize(); return time.perf_counter() - t0


one()
print('warm pass: %.1f ms / step' % (one() / 8 * 1e3))
pr = cProfile.Profile(); pr.enable(); dt = one(); pr.disable()
print('profiled pass: %.1f ms / step' % (dt / 8 * 1e3))
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(25)
