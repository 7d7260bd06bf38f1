"""cProfile of one timed RuntimeCalibrationPass (ResNet-50, KL, 8 x 32): where the host time goes."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.backends.cudnn.benchmark = True
dev = 'cuda'
g = torch.Generator().manual_seed(1)
batches = [torch.rand(32, 3, 224, 224, generator=g).to(dev) for _ in range(8)]
graph, ex = bench.build_workload(dev, 2048, 'kl')
bench.run_pass(graph, ex, batches[:1], 1, 'kl')
graph, ex = bench.build_workload(dev, 2048, 'kl')
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
bench.run_pass(graph, ex, batches, 8, 'kl', False, 'auto')
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
