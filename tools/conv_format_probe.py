"""Probe: ResNet-50 topology FP32 forward time, NCHW vs channels_last tensors (MIOpen picks different solvers)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppq_amd import harness

dev = 'cuda'
torch.backends.cudnn.benchmark = True
graph = harness.resnet50_graph(seed=0)
ex = harness.TorchExecutor(graph, dev)
x = torch.rand(32, 3, 224, 224, device=dev)


def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ex.forward(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


run(2); print('NCHW           ms/forward', round(run(10), 3))
for v in graph.variables.values():
    if v.is_parameter and v.value is not None and v.value.ndim == 4:
        v.value = v.value.contiguous(memory_format=torch.channels_last)
x = x.contiguous(memory_format=torch.channels_last)
run(2); print('channels_last  ms/forward', round(run(10), 3))
