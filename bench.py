"""bench.py -- calibration samples/s of RuntimeCalibrationPass (KL, 2048 bins) on ResNet-50 INT8.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched by
torch.distributed.run, one rank per GPU over RCCL).  A *step* is one calibration batch
(`--batch` samples, default 32) taken through BOTH phases of the pass; the timed region is exactly
one `RuntimeCalibrationPass.optimize(...)` with `calib_steps = K` (phase 1 range collection, render,
phase 2 histogram collection incl. the per-forward weight fake-quant the reference performs, batched
KL search, render), bracketed by barrier + torch.cuda.synchronize on both sides, MAX over ranks.
Inputs (K batches of torch.rand(batch,3,224,224)) are resident in HBM before the timer starts.
Weak scaling: every rank calibrates K batches of its own; statistics merge with one RCCL all-reduce
per phase (ppq_amd/distributed.py); value = N*K*batch / time.

Rank 0 prints ONE JSON line; `roofline` describes the dominant ppq_amd kernel of the timed workload
(hipEvent pairs on the launch stream, collected in a second identical pass), `cpu_baseline` is the
CPU oracle (oracle/cpu_calibration.py: torch-CPU dense ops + the C restatement of the kernels) on a
bounded sample of the same workload (N == 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def setup_dist(n_gpus: int, backend: str = 'nccl', single_device: bool = False):
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
    local = 0 if single_device else int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':     # RCCL over xGMI
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:                     # debugging aid: several ranks on one GPU (RCCL refuses duplicate devices)
            dist.init_process_group(backend)
    return rank, world, local


def barrier(world):
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


TRACE_STEPS = None


def build_workload(dev, bins, method, cache_params=False, fuse_params=True, channels_last=False):
    from ppq_amd import harness
    graph = harness.resnet50_graph(seed=0)
    harness.quantize_graph(graph, method, hist_bins=bins)
    ex = harness.TorchExecutor(graph, dev)
    ex.cache_parameter_quantization = bool(cache_params)
    ex.fuse_parameter_quantization = bool(fuse_params)
    if channels_last: ex.use_channels_last()
    harness.ParameterQuantizePass().optimize(graph)        # weights: per-channel min-max, left ACTIVATED
    return graph, ex


def run_pass(graph, ex, batches, steps, method, async_observe=False, hip_graph=False, batch_observations=True,
             reuse_activations=False, queue_bytes=None):
    from ppq_amd.calibration import RuntimeCalibrationPass
    p = RuntimeCalibrationPass(method=method, check_steps=False, async_observe=async_observe, use_hip_graph=hip_graph,
                               batch_observations=batch_observations, reuse_activations=reuse_activations,
                               queue_bytes=queue_bytes)
    if TRACE_STEPS is not None:          # --trace-steps: host timestamp + device event after every forward
        inner = p._forward

        def traced(executor, data, hooks, output_names):
            inner(executor, data, hooks, output_names)
            ev = torch.cuda.Event(enable_timing=True); ev.record()
            TRACE_STEPS.append((time.perf_counter(), ev))
        p._forward = traced
        inner_render = p._render

        def traced_render():
            torch.cuda.synchronize(); a = time.perf_counter()
            inner_render()
            torch.cuda.synchronize(); TRACE_STEPS.append((a, time.perf_counter()))
        p._render = traced_render
    p.optimize(graph, dataloader=batches, executor=ex, calib_steps=steps)
    return p


def collect_prof():
    from ppq_amd import _lib
    arr = (_lib.ProfEntry * 32)()
    n = _lib.lib.ppqhip_prof_collect(arr, 32)
    return [{'name': arr[i].name.decode(), 'launches': int(arr[i].launches), 'total_ms': float(arr[i].total_ms),
             'total_bytes': float(arr[i].total_bytes)} for i in range(n)]


def pmc_traffic(kernel_name: str):
    """HBM bytes per launch of `kernel_name` from the committed PMC summary (separate rocprofv3
    `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of THIS command, FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md; tools/summarize_profile.py).  None when no summary is present."""
    path = os.path.join(ROOT, 'profiles', 'r01_bench_pmc_bytes.json')
    if not os.path.exists(path): return None, None
    data = json.load(open(path))
    keys = {'hist_sym_t': ('hist_t_multi_kernel<false', 'hist_t_lds_kernel<false'),
            'hist_asym_t': ('hist_t_multi_kernel<true', 'hist_t_lds_kernel<true'),
            'minmax_t': ('minmax_t_multi_kernel', 'minmax_t_kernel'),
            'fq_linear_c': ('fq_linear_multi_kernel', 'fq_linear_c_tile_kernel'),
            'fq_linear_t': ('fq_linear_t_tile_kernel',)}.get(kernel_name)
    if keys is None: return None, None
    tot = n = 0
    for k, v in data.items():
        if k.startswith(keys) and 'launches' in v:
            tot += v['launches'] * (v.get('hbm_read_bytes_per_launch_corrected', 0) + v.get('hbm_write_bytes_per_launch', 0))
            n += v['launches']
    return (round(tot / n), 'profiles/r01_bench_pmc_bytes.json') if n else (None, None)


def cpu_baseline(bins, batch_samples=2, n_batches=4):
    """The same two-phase KL calibration on the host cores, through the CPU oracle."""
    from ppq_amd import harness
    from oracle.cpu_calibration import timed_calibrate_cpu
    graph = harness.resnet50_graph(seed=0)
    harness.quantize_graph(graph, 'kl', hist_bins=bins)
    g = torch.Generator().manual_seed(1)
    batches = [torch.rand(batch_samples, 3, 224, 224, generator=g) for _ in range(n_batches)]
    secs, _ = timed_calibrate_cpu(graph, batches, bins)
    n = batch_samples * n_batches
    return {'value': round(n / secs, 3), 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n} samples ({n_batches} batches of {batch_samples}) of the same ResNet-50 KL calibration: '
                      f'torch-CPU dense ops on {torch.get_num_threads()} threads + single-threaded C oracle kernels; '
                      f'{secs:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--bins', type=int, default=2048)
    ap.add_argument('--method', type=str, default='kl')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', type=str, default='nccl')
    ap.add_argument('--single-device', type=int, default=0, help='debug: put every rank on cuda:0 (use with --backend gloo)')
    ap.add_argument('--hip-graph', default='auto', choices=['0', '1', 'auto'],
                    help="replay each phase's forward as a HIP graph: never / always / when one timed eager step is launch-bound")
    ap.add_argument('--async-observe', type=int, default=0, help='observer kernels on a side HIP stream')
    ap.add_argument('--fuse-params', type=int, default=1, help='all weights of a forward fake-quantised by one multi-tensor launch')
    ap.add_argument('--batch-observations', type=int, default=1, help='all statistics kernels of a forward in one multi-tensor launch')
    ap.add_argument('--settle-ms', type=float, default=400.0, help='untimed: extra warm-up forwards (ms of wall clock) so device clocks settle')
    ap.add_argument('--trace-steps', type=int, default=0, help='debug: per-forward host / device timestamps in the JSON line')
    ap.add_argument('--reuse-activations', type=int, default=0,
                    help='OPT-IN, off for the headline number: keep the phase-1 activations in HBM and bin them in phase 2 instead of running the forward again')
    ap.add_argument('--queue-mib', type=int, default=0, help='debug: ObservationQueue flush threshold (MiB), 0 = default')
    ap.add_argument('--channels-last', type=int, default=0, help='activations and conv weights in channels-last (NHWC) memory format')
    ap.add_argument('--cache-params', type=int, default=0, help='keep fake-quantised weights resident between forwards')
    ap.add_argument('--miopen-find', type=int, default=1, help='torch.backends.cudnn.benchmark (MIOpen find mode)')
    args = ap.parse_args()

    rank, world, local = setup_dist(args.gpus, args.backend, bool(args.single_device))
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    dev = f'cuda:{local}'
    import ppq_amd  # noqa: F401  (fails loudly without libppq_hip.so)
    from ppq_amd import _lib

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    batches = [torch.rand(args.batch, 3, 224, 224, device=dev, generator=g) for _ in range(args.steps)]

    # warm-up: W batches through a complete two-phase pass (MIOpen find, library load, allocator)
    if args.warmup > 0:
        graph, ex = build_workload(dev, args.bins, args.method, args.cache_params, args.fuse_params, bool(args.channels_last))
        run_pass(graph, ex, batches[: max(1, min(args.warmup, args.steps))], max(1, min(args.warmup, args.steps)), args.method,
                 bool(args.async_observe), False, bool(args.batch_observations))
        # keep the device under the workload's own load profile for a moment: a GPU that sat idle (fresh
        # box) otherwise spends the first timed steps in clock / power transitions (sporadic 10-20 ms
        # device-side stalls were traced to the first steps after idle; tools/find_stall.py)
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for b in batches[:2]: ex.forward(b)
            torch.cuda.synchronize()
        del graph, ex

    # timed region
    global TRACE_STEPS
    if args.trace_steps:
        TRACE_STEPS = []
        ev0 = torch.cuda.Event(enable_timing=True)
    graph, ex = build_workload(dev, args.bins, args.method, args.cache_params, args.fuse_params, bool(args.channels_last))
    barrier(world)
    t0 = time.perf_counter()
    if args.trace_steps: ev0.record()
    p = run_pass(graph, ex, batches, args.steps, args.method, bool(args.async_observe), {'0': False, '1': True, 'auto': 'auto'}[args.hip_graph],
                 bool(args.batch_observations), bool(args.reuse_activations), (args.queue_mib << 20) or None)
    barrier(world)
    elapsed = time.perf_counter() - t0
    trace = None
    if args.trace_steps:
        trace = [{'host_ms': round((t - t0) * 1e3, 2), 'dev_ms': round(ev0.elapsed_time(e), 2)} if not isinstance(e, float)
                 else {'render_start_ms': round((t - t0) * 1e3, 2), 'render_ms': round((e - t) * 1e3, 2)} for t, e in TRACE_STEPS]
        TRACE_STEPS = None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_obs = sum(len(o.observers()) for o in p._observers.values())
    scale_checksum = float(sum(float(c.scale.sum()) for op in graph.operations.values() if hasattr(op, 'config')
                               for c, v in op.config_with_variable if not v.is_parameter and c.scale is not None
                               and int(getattr(c.state, 'value', c.state)) == 4))

    # roofline leg: the identical pass once more with hipEvent pairs around every library launch
    roof = None
    prof_rows = []
    if rank == 0:
        graph2, ex2 = build_workload(dev, args.bins, args.method, args.cache_params, args.fuse_params, bool(args.channels_last))
        torch.cuda.synchronize()
        _lib.lib.ppqhip_prof_enable(1)
        if world == 1:
            run_pass(graph2, ex2, batches, args.steps, args.method, False, False, bool(args.batch_observations), False,
                     (args.queue_mib << 20) or None)   # eager, one stream -> clean event pairs
        else:   # collectives need every rank; profile the local (non-merged) statistics path only
            from ppq_amd.calibration import RuntimeCalibrationPass
            pp = RuntimeCalibrationPass(method=args.method, check_steps=False, async_observe=False, use_hip_graph=False)
            pp._render = lambda: __import__('ppq_amd.observer', fromlist=['render_observers']).render_observers(pp._all_tensor_observers())
            pp.optimize(graph2, dataloader=batches, executor=ex2, calib_steps=args.steps)
        torch.cuda.synchronize()
        _lib.lib.ppqhip_prof_enable(0)
        prof_rows = collect_prof()
        if prof_rows:
            # An event pair reports kernel duration + the command processor's timestamp / dispatch
            # overhead; the library measures that overhead with EMPTY pairs on the same stream and it is
            # subtracted, so avg_launch_us is comparable with rocprofv3's begin->end kernel duration
            # (profiles/r01_bench_kernel_stats.csv).  The raw pair time is reported next to it.
            _lib.lib.ppqhip_prof_event_overhead_us(torch.cuda.current_stream().cuda_stream, 16)      # warm
            overhead_us = max(0.0, float(_lib.lib.ppqhip_prof_event_overhead_us(torch.cuda.current_stream().cuda_stream, 256)))
            dom = max(prof_rows, key=lambda r: r['total_ms'])
            raw_s = dom['total_ms'] * 1e-3 / dom['launches']
            avg_s = max(raw_s - overhead_us * 1e-6, 0.25 * raw_s)
            avg_b = dom['total_bytes'] / dom['launches']
            ach = avg_b / avg_s / 1e9
            traffic, traffic_src = pmc_traffic(dom['name'])
            roof = {'kernel': dom['name'], 'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBPS,
                    'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBPS, 4), 'traffic': traffic,
                    'traffic_source': traffic_src,
                    'launches': dom['launches'], 'avg_launch_us': round(avg_s * 1e6, 2),
                    'avg_event_pair_us': round(raw_s * 1e6, 2), 'event_overhead_us': round(overhead_us, 2),
                    'algorithmic_bytes_per_launch': round(avg_b)}
    if world > 1:
        barrier(world)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.bins)

    if rank == 0:
        samples = world * args.steps * args.batch
        out = {
            'metric': ('calibration samples/sec (RuntimeCalibrationPass, KL 2048-bin, ResNet-50 INT8)' if args.method == 'kl' and args.bins == 2048
                       else f'calibration samples/sec (RuntimeCalibrationPass, {args.method}, {args.bins} bins, ResNet-50 INT8)'),
            'value': round(samples / elapsed, 2), 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'ResNet-50 topology (53 Conv + 1 Gemm, BN folded, seeded He init), '
                                   f'RuntimeCalibrationPass {args.method} {args.bins} bins, per-tensor INT8 activations, '
                                   f'per-channel INT8 weights, {args.steps} batches x {args.batch} x 3x224x224 per GPU',
                       'samples': samples, 'batch': args.batch, 'observed_tensors': n_obs,
                       'parallelism': f'dp{world} (batches sharded, 1 all-reduce per phase)',
                       'async_observe': bool(args.async_observe), 'cache_params': bool(args.cache_params),
                       'channels_last': bool(args.channels_last), 'fuse_params': bool(args.fuse_params), 'batch_observations': bool(args.batch_observations),
                       'reuse_activations': bool(args.reuse_activations), 'replayed_batches': p.replayed_batches,
                       'hip_graph': args.hip_graph, 'graph_replays': p.graph_replays,
                       'graph_decisions': [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items()}
                                           for d in p.graph_decisions]},
            'roofline': roof, 'cpu_baseline': cpu,
            'kernels': [{'name': r['name'], 'launches': r['launches'], 'total_ms': round(r['total_ms'], 3),
                         'GBps': round(r['total_bytes'] / max(r['total_ms'], 1e-9) / 1e6, 1)} for r in prof_rows],
            'scale_checksum': scale_checksum,
        }
        if trace is not None: out['trace_steps'] = trace
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
