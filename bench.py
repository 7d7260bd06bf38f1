"""bench.py -- calibration samples/s of RuntimeCalibrationPass (KL, 2048 bins) on ResNet-50 INT8.

Contract: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the ranks are either launched
by torch.distributed.run (RANK / WORLD_SIZE in the environment) or -- when bench.py is started plainly
with --gpus N -- spawned by bench.py itself (it re-executes itself under torch.distributed.run on
127.0.0.1), one rank per GPU over RCCL.  A *step* is one calibration batch
(`--batch` samples, default 32) taken through BOTH phases of the pass; the timed region is exactly
one `RuntimeCalibrationPass.optimize(...)` with `calib_steps = K` (phase 1 range collection, render,
phase 2 histogram collection incl. the per-forward weight fake-quant the reference performs, batched
KL search, render), bracketed by barrier + torch.cuda.synchronize on both sides, MAX over ranks.
Inputs (K batches of torch.rand(batch,3,224,224)) are resident in HBM before the timer starts.
Weak scaling: every rank calibrates K batches of its own; statistics merge with two RCCL collectives per
phase -- a 13-word layout probe (MAX) and ONE data all-reduce over one flat buffer (ppq_amd/distributed.py);
value = N*K*batch / time.  Strong scaling (`--total-samples S`, e.g. BASELINE config 3's "1024 samples sharded across 8"): the job is
fixed, K = S / (N * batch) is derived, `scaling` is "strong"; each rank of a multi-GPU run keeps its own MIOpen user database /
kernel cache so that N find-mode warm-ups do not serialise on one sqlite file.

The timed pass is repeated `--repeats` times (default 5, each on a freshly built graph, each timing
exactly K steps); `value` is the MEDIAN pass and `values` / `spread_pct` report all of them.

Rank 0 prints ONE JSON line; `roofline` describes the dominant ppq_amd kernel of the timed workload
(hipEvent pairs on the launch stream, collected in one more identical pass; `traffic` = HBM bytes per
launch from rocprofv3 PMC passes of this very command run as child processes, FETCH_SIZE and WRITE_SIZE
separately, gfx950 correction applied -- null when rocprofv3 is unavailable), `cpu_baseline` is PPQ's
USING_CUDA_KERNEL=False PyTorch-CPU path (oracle/torch_cpu_path.py, or the reference itself when a staged
copy is importable) on a bounded sample of the same workload, `cpu_ops` the per-op table of BASELINE.md
section 3 (N == 1 only).

Because the driver keeps the SCALAR keys of `config` and drops nested ones, every figure of `config.variants` that matters is
repeated as a flat scalar of `config` (N == 1, default run): `batch1_x256_samples_per_s`, `seam_{kernels,pass}_b{32,1}_samples_per_s`,
`cfg3 / cfg4 / cfg5_samples_per_s` (+ their `_roofline_frac`, `_spread_pct`), `cfg5_replay_ms_per_step`, `kl4096_`, `percentile_`,
`reuse_activations_samples_per_s`, and the north star's own tensor (tools/north_star.py): `B_<kernel>_rocprof_median_us` / `_p10_us` /
`_frac_of_8TBps` / `_over_floor` = the MEDIAN device duration of the single-tensor launch on B = [1,512,56,56] from a `rocprofv3
--kernel-trace` child, its fraction of 8 TB/s and its ratio to the matching FLOOR kernel measured in the same child (`B_floor_empty_us`,
`B_floor_read_us`, `B_floor_copy_us`, `B_floor_read_atomic_*_us`); `B_<kernel>_graph_us` = the same launch replayed 200x from one HIP graph /
200 in THIS process (no profiler needed: an upper bound that exists on any box); `<kernel>_frac_x{2,4,8,16,32}` and
`<kernel>_crosses_0p70_at_x` = the same launch on B x k and the smallest measured k at which it reaches 0.70 of 8 TB/s; `B_status` says
whether the rocprofv3 child worked and, if not, why (stderr tail).  `roofline.rocprof_median_us` / `rocprof_frac` = the dominant kernel's
median duration from a clean `rocprofv3 --kernel-trace` child pass of this command, next to the event-timed `avg_launch_us`.

`config.variants` (N == 1, default run) also carries, each timed on this box in this run:
  * SURVEY 8(d)'s own protocol (batch 1 x 256 steps) with its wall time decomposed by phase / render and the eager step's
    host-issue vs device time;
  * the same pass with `reuse_activations`;
  * the two PLUGIN SEAMS inside the unmodified reference (when a staged copy is importable): reference executor + pass +
    observers on these kernels (`install_into_ppq`), and reference executor driving THIS package's pass + observers
    (`install_plugins_into_ppq`) -- the reference is the host there, what is measured is this library underneath it;
  * BASELINE configs 3, 4 and 5 as short child runs of this file (`--workload resnet50_cfg3 | vit_b16_fp8 |
    yolov6s_int4_lsq`), each with the roofline entry of ITS dominant kernel (hist_asym_t, fq_float_*, lsq_bwd_*); a child warms
    itself with MIOpen's find mode on, times 8 steps three times and reports the spread.
"""
import argparse
import csv
import glob
import gc
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time



def host_cpus() -> int:
    """CPUs this process can really keep busy: its affinity mask cut by the container's CPU quota (cgroup v2 cpu.max, v1
    cfs_quota_us).  The GPU boxes of this pool show 256 CPUs and grant 16: a default-sized thread pool (128 OpenMP threads spinning
    behind every parallel region) spends the quota in a few milliseconds and the kernel then parks EVERY thread of the container --
    the one that launches kernels too -- until the next 100 ms period: whole timed passes came out 75-85 ms long (one in five)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f: quota, period = f.read().split()[:2]
        if quota != 'max': n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f: quota = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f: period = int(f.read())
            if quota > 0: n = min(n, max(1, quota // period))
        except (OSError, ValueError): pass
    return max(1, n)


def cgroup_cpu_stat() -> dict:
    """usage_usec / nr_throttled / throttled_usec of this container (cgroup v2), {} where there is no such file."""
    try:
        with open('/sys/fs/cgroup/cpu.stat') as f: return {k: int(v) for k, v in (line.split() for line in f)}
    except (OSError, ValueError): return {}


HOST_CPUS = host_cpus()
_RANK_CPUS = max(1, HOST_CPUS // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1'))))     # the ranks of one node share the grant
for _var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):      # before torch / numpy build their pools
    os.environ.setdefault(_var, str(_RANK_CPUS))
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')                             # idle workers sleep instead of spinning the quota away

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

JSON_OUT = sys.stdout        # the real stdout, kept for the one JSON line (see __main__)
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def maybe_spawn(args) -> None:
    """`python bench.py --gpus N` from a plain shell: run the N ranks ourselves (one per GPU) by
    re-executing this file under torch.distributed.run on the loopback interface."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ or 'RANK' in os.environ: return
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0)); port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL / cross-process sharing needs it here
    env['OMP_NUM_THREADS'] = str(max(1, HOST_CPUS // args.gpus))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def per_rank_miopen_dirs() -> None:
    """Every rank of a multi-GPU run warms its convolutions with MIOpen's find mode; with ONE user database / kernel cache the eight
    searches serialise on one sqlite file (and lock-wait each other).  Give each rank its own directories, before MIOpen initialises
    (whoever launched the ranks: torch.distributed.run from the driver, or maybe_spawn).  Single-rank runs keep the defaults."""
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1: return
    r = os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0'))
    base = os.path.join(tempfile.gettempdir(), f'ppq_miopen_{os.getuid()}_rank{r}')
    for var, sub in (('MIOPEN_USER_DB_PATH', 'db'), ('MIOPEN_CUSTOM_CACHE_DIR', 'cache')):
        if var not in os.environ:
            os.makedirs(os.path.join(base, sub), exist_ok=True)
            os.environ[var] = os.path.join(base, sub)


def setup_dist(n_gpus: int, backend: str = 'nccl', single_device: bool = False):
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
    if world != max(1, n_gpus):
        raise SystemExit(f'bench.py: --gpus {n_gpus} but WORLD_SIZE={world}: launch one rank per GPU '
                         f'(or start bench.py plainly and let it spawn the ranks)')
    local = 0 if single_device else int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':     # RCCL over xGMI
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:                     # debugging aid: several ranks on one GPU (RCCL refuses duplicate devices)
            dist.init_process_group(backend)
    return rank, world, local


def host_selftest(args) -> None:
    """`--host-selftest`: the multi-process plumbing WITHOUT a GPU -- spawn, rendezvous on 127.0.0.1, the flat
    per-phase merge of ppq_amd.distributed over `gloo` on CPU buffers shaped like the ResNet-50 statistics
    (72 ranges, 72 x bins int32 histograms), MAX-over-ranks timing, one JSON line from rank 0.  What the CPU
    suite runs (tests/test_host_cpu.py); the data path itself needs the GPU."""
    import torch.distributed as dist
    from ppq_amd.distributed import last_merge_stats, merge_observers  # noqa: F401
    from ppq_amd import distributed as D
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if world != max(1, args.gpus): raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if world > 1: dist.init_process_group('gloo')

    class Stat:
        def __init__(self, bufs): self.bufs = bufs
        def reducible(self): return self.bufs
    g = torch.Generator().manual_seed(99 + rank)
    rngs = [torch.stack([-torch.rand(1, generator=g), torch.rand(1, generator=g)]).reshape(2) for _ in range(72)]
    hists = [torch.randint(0, 1000, [args.bins], generator=g, dtype=torch.int32) for _ in range(72)]
    total_before = sum(int(h.sum()) for h in hists)
    t0 = time.perf_counter()
    merge_observers([Stat([(r[0:1], 'min'), (r[1:2], 'max')]) for r in rngs])
    phase1 = dict(D.last_merge_stats)
    merge_observers([Stat([(h, 'sum')]) for h in hists])
    phase2 = dict(D.last_merge_stats)
    elapsed = time.perf_counter() - t0
    ok = True
    if world > 1:
        t = torch.tensor([elapsed, float(total_before)], dtype=torch.float64)
        m = t.clone(); dist.all_reduce(m, op=dist.ReduceOp.MAX); elapsed = float(m[0])
        tot = t[1:2].clone(); dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        ok = int(tot[0]) == sum(int(h.sum()) for h in hists)          # merged histograms == sum of the shards
    if rank == 0:
        print(json.dumps({'metric': 'host selftest of the data-parallel merge (no GPU work)', 'value': round(elapsed * 1e3, 3),
                          'unit': 'ms', 'n_gpus': world, 'rccl_ranks': world, 'backend': 'gloo', 'selftest': True,
                          'merged_counts_ok': ok,
                          'merge': [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in m.items()} for m in (phase1, phase2)]}),
              file=JSON_OUT, flush=True)
    if world > 1: dist.destroy_process_group()


def barrier(world):
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


TRACE_STEPS = None


WORKLOAD = 'resnet50'      # --workload: which BASELINE.json config the line measures (default: config 2, the metric's own)
WORKLOADS = {
    # name: (description, calibration method or None = the args' / the quantizer's own)
    'resnet50': ('ResNet-50 topology (53 Conv + 1 Gemm, BN folded, seeded He init), per-tensor INT8 activations, per-channel INT8 weights', None),
    'resnet50_cfg3': ('BASELINE config 3: ResNet-50 topology, per-channel ASYMMETRIC INT8 weights + per-tensor asymmetric activations, MSE clipping search', 'mse'),
    'vit_b16_fp8': ('BASELINE config 4: ViT-B/16 topology (86.6 M parameters), TRT_FP8 policy: FP8 E4M3 inputs of Conv / Gemm / MatMul, power-of-2 scales from the floating observer', 'floating'),
    'yolov6s_int4_lsq': ('BASELINE config 5: YOLOv6-s-like detector (56 Conv, 17 M parameters, 6 outputs), INT4 per-channel weights + INT8 activations, '
                         'block-wise LearnedStepSizePass (block_size 5: 27 blocks) through the HIP forward / LSQ-backward kernels', 'minmax'),
}


def build_workload(dev, bins, method, cache_params=False, fuse_params=True, channels_last=False):
    from ppq_amd import harness
    if WORKLOAD == 'vit_b16_fp8':
        graph = harness.vit_graph(seed=0)
        harness.quantize_graph_fp8(graph)
    elif WORKLOAD == 'resnet50_cfg3':
        graph = harness.resnet50_graph(seed=0)
        harness.quantize_graph(graph, method, symmetrical=False, weight_symmetrical=False, hist_bins=bins)
    else:
        graph = harness.resnet50_graph(seed=0)
        harness.quantize_graph(graph, method, hist_bins=bins)
    ex = harness.TorchExecutor(graph, dev)
    ex.cache_parameter_quantization = bool(cache_params)
    ex.fuse_parameter_quantization = bool(fuse_params)
    if channels_last: ex.use_channels_last()
    harness.ParameterQuantizePass().optimize(graph)        # weights: per-channel min-max, left ACTIVATED
    return graph, ex


def run_pass(graph, ex, batches, steps, method, async_observe=False, hip_graph=False, batch_observations=True,
             reuse_activations=False, queue_bytes=None, decompose=None):
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
    if decompose is not None:            # wall time of each forward loop and each render (synchronised on both sides)
        inner_cal, inner_ren = p.calibrate, p._render

        def cal(**kw):
            torch.cuda.synchronize(); a = time.perf_counter()
            inner_cal(**kw)
            torch.cuda.synchronize(); decompose.setdefault('forward_loops_ms', []).append(round((time.perf_counter() - a) * 1e3, 2))

        def ren():
            torch.cuda.synchronize(); a = time.perf_counter()
            inner_ren()
            torch.cuda.synchronize(); decompose.setdefault('renders_ms', []).append(round((time.perf_counter() - a) * 1e3, 2))
        p.calibrate, p._render = cal, ren
    p.optimize(graph, dataloader=batches, executor=ex, calib_steps=steps)
    return p


def collect_prof():
    from ppq_amd import _lib
    arr = (_lib.ProfEntry * 32)()
    n = _lib.lib.ppqhip_prof_collect(arr, 32)
    return [{'name': arr[i].name.decode(), 'launches': int(arr[i].launches), 'total_ms': float(arr[i].total_ms),
             'total_bytes': float(arr[i].total_bytes)} for i in range(n)]


# device kernels that implement each logical library kernel (prefixes of the demangled names rocprofv3 prints)
DEVICE_KERNELS = {'hist_sym_t': ('ppqhip::hist_persistent_kernel<false', 'ppqhip::hist_small_kernel<false'),
                  'hist_asym_t': ('ppqhip::hist_persistent_kernel<true', 'ppqhip::hist_small_kernel<true'),
                  'minmax_t': ('ppqhip::minmax_persistent_kernel', 'ppqhip::minmax_small_kernel', 'ppqhip::minmax_t_kernel'),
                  'fq_linear_c': ('ppqhip::fq_linear_multi_kernel', 'ppqhip::fq_linear_c_tile_kernel'),
                  'fq_linear_t': ('ppqhip::fq_linear_t_tile_kernel',),
                  # a quantile_t "launch" is one 7-kernel sequence: bytes of ALL its kernels (init zeroes 84 KB per job, the
                  # filter reads the tensors, select A reads the lists ..) per filter launch (one per sequence)
                  'quantile_t': ('ppqhip::quantile_',)}
LAUNCHES_COUNTED_ON = {'quantile_t': 'ppqhip::quantile_filter_kernel'}


def pmc_traffic(kernel_name: str, child_args: list, timeout_s: float = 240.0):
    """HBM bytes per launch of the device kernel(s) behind `kernel_name`, MEASURED NOW: this command is
    run twice more as a child under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE`
    (separate passes, nothing but kernel tracing next to the counters), a reduced step count, no
    baselines.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies the 128-B requests of a
    wide coalesced stream at 64 B -> doubled; WRITE_SIZE is taken as reported (KB).  Returns
    (bytes per launch | None, note)."""
    prefixes = DEVICE_KERNELS.get(kernel_name)
    rocprof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if prefixes is None or rocprof is None: return None, 'rocprofv3 not available'
    per_launch = {}
    work = tempfile.mkdtemp(prefix='ppq_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(work, counter)
            cmd = [rocprof, '--output-format', 'csv', '--kernel-trace', '--pmc', counter, '-d', out, '-o', 'pmc', '--',
                   sys.executable, os.path.abspath(__file__)] + child_args
            try:
                subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            except (subprocess.TimeoutExpired, OSError) as e:
                return None, f'{counter} pass failed: {type(e).__name__}'
            files = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
            if not files: return None, f'{counter} pass wrote no counter_collection.csv'
            tot = n = 0
            counted_on = LAUNCHES_COUNTED_ON.get(kernel_name)
            for r in csv.DictReader(open(files[0])):
                name = r['Kernel_Name'].replace('void ', '')
                if r['Counter_Name'] == counter and name.startswith(prefixes):
                    tot += float(r['Counter_Value'])
                    if counted_on is None or name.startswith(counted_on): n += 1
            if n == 0: return None, f'{counter}: kernel not found in the trace'
            per_launch[counter] = tot / n * 1024.0        # the counters report KB
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return round(2.0 * per_launch['FETCH_SIZE'] + per_launch['WRITE_SIZE']), \
        'live rocprofv3 --pmc passes of this command (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE)'


def staged_reference():
    """A staged, importable copy of the reference (tools/stage_reference.py -> oracle/_ref/ppq_stage, git-ignored):
    returns its path or None.  /root/reference itself is never read here."""
    path = os.path.join(ROOT, 'oracle', '_ref', 'ppq_stage')
    return path if os.path.isdir(os.path.join(path, 'ppq')) else None


def cpu_baseline(bins, target_samples=32, batch_samples=4, budget_s=15.0):
    """PPQ's USING_CUDA_KERNEL=False path on the host cores: the same two-phase KL calibration of the same
    ResNet-50 graph, torch-CPU ops on all host threads (torch.histc / min / max / round, range.py:86-98,
    175-188, qfunction/linear.py:27-32).  The reference's own classes when a staged copy is importable
    (kind "reference"), otherwise their line-by-line restatement oracle/torch_cpu_path.py ("reference-path").
    The single-threaded C oracle (round 1's baseline) is reported next to it on a smaller sample."""
    from ppq_amd import harness
    from oracle import torch_cpu_path as T
    from oracle.cpu_calibration import timed_calibrate_cpu
    g = torch.Generator().manual_seed(1)
    kind, runner = 'reference-path', T.timed_calibrate
    stage = staged_reference()
    if stage is not None:
        try:
            from oracle import reference_import as RI
            RI.load(stage)
            kind, runner = 'reference', RI.timed_calibrate
        except Exception as e:        # a broken stage must not cost the bench line
            print(f'[bench] staged reference not usable ({e}); timing the restated path', file=sys.stderr)

    def graph():
        gr = harness.resnet50_graph(seed=0)
        harness.quantize_graph(gr, 'kl', hist_bins=bins)
        return gr
    # one warm batch sizes the sample: ~budget_s of host work, at least target_samples when that fits
    warm = [torch.rand(batch_samples, 3, 224, 224, generator=g)]
    secs1, _ = runner(graph(), warm, bins)
    n_batches = max(1, min(target_samples // batch_samples, int(budget_s / max(secs1, 1e-3))))
    batches = [torch.rand(batch_samples, 3, 224, 224, generator=g) for _ in range(n_batches)]
    runs = [runner(graph(), batches, bins)[0] for _ in range(2)]        # best of two: the first pays page faults / thread start-up
    secs = min(runs)
    n = batch_samples * n_batches
    small = [torch.rand(2, 3, 224, 224, generator=g) for _ in range(2)]
    psecs, _ = timed_calibrate_cpu(graph(), small, bins)
    return {'value': round(n / secs, 3), 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': kind,
            'sample': f'{n} samples ({n_batches} batches of {batch_samples}) of the same ResNet-50 KL-{bins} calibration, '
                      f"PPQ's USING_CUDA_KERNEL=False torch-CPU path on {torch.get_num_threads()} threads; best of 2 runs: "
                      + ' / '.join(f'{t:.1f} s' for t in runs),
            'values': [round(n / t, 3) for t in runs],
            'port_c_oracle': {'value': round(4 / psecs, 3), 'unit': 'samples/s', 'cores': 1,
                              'sample': f'4 samples, torch-CPU dense ops + single-threaded C restatement of the kernels; {psecs:.1f} s'}}


def roofline_entry(prof_rows, prefer=None):
    """The `roofline` object for the dominant library kernel of `prof_rows` (or the first row whose name starts with one of
    `prefer`).  An event pair reports kernel duration + the command processor's timestamp / dispatch overhead; the library
    measures that overhead with EMPTY pairs on the same stream and it is subtracted, so avg_launch_us is comparable with
    rocprofv3's begin->end kernel duration.  The raw pair time is reported next to it."""
    from ppq_amd import _lib
    if not prof_rows: return None
    _lib.lib.ppqhip_prof_event_overhead_us(torch.cuda.current_stream().cuda_stream, 16)      # warm
    overhead_us = max(0.0, float(_lib.lib.ppqhip_prof_event_overhead_us(torch.cuda.current_stream().cuda_stream, 256)))
    rows = [r for r in prof_rows if prefer and r['name'].startswith(tuple(prefer))] or prof_rows
    dom = max(rows, key=lambda r: r['total_ms'])
    raw_s = dom['total_ms'] * 1e-3 / dom['launches']
    avg_s = max(raw_s - overhead_us * 1e-6, 0.25 * raw_s)
    avg_b = dom['total_bytes'] / dom['launches']
    ach = avg_b / avg_s / 1e9
    return {'kernel': dom['name'], 'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBPS,
            'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBPS, 4), 'traffic': None, 'traffic_source': None,
            'launches': dom['launches'], 'avg_launch_us': round(avg_s * 1e6, 2),
            'avg_event_pair_us': round(raw_s * 1e6, 2), 'event_overhead_us': round(overhead_us, 2),
            'algorithmic_bytes_per_launch': round(avg_b),
            'frac_of_measured_copy_ceiling': round(ach / 6290.0, 4)}      # MI355X_MICROARCH.md: 6.29 TB/s float4 copy


def north_star_b(bins):
    """The tensor BASELINE.json's north star names, B = [1,512,56,56] fp32 (6.4 MB), and B x {2..32}, through the single-tensor entry
    points, next to FLOOR kernels (empty / pure read / copy / read + atomic combine) measured the same way: tools/north_star.py.
    Method 1: MEDIAN device durations from `rocprofv3 --kernel-trace` of a child (these launches are a few microseconds long and
    latency-bound: event pairs cannot time them, their own overhead is as long as the kernel).  Method 2 (always, in this process,
    no profiler): HIP-graph replay of back-to-back launches / count -- an upper bound that exists on any box.  Returns flat scalars
    incl. `B_status` (why method 1 failed, when it did: BENCH_r05's line silently lost these keys)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import north_star
        m = north_star.measure(bins)
        scal = north_star.flat_scalars(m)
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        try: north_star.report(m, os.path.join(ROOT, 'gpurun_out', 'frac_vs_size.txt'))
        except Exception: pass
        return scal
    except Exception as e:      # never costs the bench line; the status key says why
        return {'B_status': f'north_star failed: {type(e).__name__}: {e}', 'B_method': None}


def rocprof_kernel_median(kernel_name: str, child_args: list, timeout_s: float = 240.0):
    """MEDIAN begin->end device duration [us] of the device kernel(s) behind `kernel_name` from a clean `rocprofv3 --kernel-trace`
    child pass of this command (no counters): what the event-timed `avg_launch_us` is checked against.  (median, n, note)."""
    prefixes = DEVICE_KERNELS.get(kernel_name)
    rocprof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if prefixes is None or rocprof is None: return None, 0, 'rocprofv3 not available'
    work = tempfile.mkdtemp(prefix='ppq_kt_', dir='/tmp')
    try:
        cmd = [rocprof, '--output-format', 'csv', '--kernel-trace', '-d', work, '-o', 'kt', '--', sys.executable, os.path.abspath(__file__)] + child_args
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), capture_output=True, text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return None, 0, f'timeout after {timeout_s:.0f} s'
        files = glob.glob(os.path.join(work, '**', '*kernel_trace.csv'), recursive=True)
        if not files: return None, 0, f'no kernel_trace.csv (rc={r.returncode}): {(r.stderr or "")[-200:]!r}'
        counted_on = LAUNCHES_COUNTED_ON.get(kernel_name)
        per_launch, cur = [], 0
        rows = sorted(csv.DictReader(open(files[0])), key=lambda q: int(q['Start_Timestamp']))
        for q in rows:                                # a logical launch = consecutive device kernels of the family (hist + reduce ..)
            name = q['Kernel_Name'].replace('void ', '')
            if not name.startswith(prefixes): continue
            d = int(q['End_Timestamp']) - int(q['Start_Timestamp'])
            if counted_on is None: per_launch.append(d)
            elif name.startswith(counted_on): per_launch.append(cur + d); cur = 0
            else: cur += d
        if not per_launch: return None, 0, 'kernel not found in the trace'
        per_launch.sort()
        big = [d for d in per_launch if d >= 0.25 * per_launch[-1]]       # the multi-tensor launches of the timed forwards, not warm-up crumbs
        return round(big[len(big) // 2] / 1e3, 2), len(big), 'clean rocprofv3 --kernel-trace child pass of this command (no counters)'
    except Exception as e:
        return None, 0, f'{type(e).__name__}: {e}'
    finally:
        shutil.rmtree(work, ignore_errors=True)


def count_dispatches(child_args: list, timeout_s: float = 240.0):
    """Kernel dispatches of a `rocprofv3 --kernel-trace` child of this file (every kernel: the library's, torch's, MIOpen's)."""
    rocprof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if rocprof is None: return None, 'rocprofv3 not available'
    work = tempfile.mkdtemp(prefix='ppq_kd_', dir='/tmp')
    try:
        cmd = [rocprof, '--output-format', 'csv', '--kernel-trace', '-d', work, '-o', 'kd', '--', sys.executable, os.path.abspath(__file__)] + child_args
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), capture_output=True, text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return None, f'timeout after {timeout_s:.0f} s'
        files = glob.glob(os.path.join(work, '**', '*kernel_trace.csv'), recursive=True)
        if not files: return None, f'no kernel_trace.csv (rc={r.returncode}): {(r.stderr or "")[-200:]!r}'
        with open(files[0]) as f: return sum(1 for _ in csv.DictReader(f)), 'ok'
    except Exception as e:
        return None, f'{type(e).__name__}: {e}'
    finally:
        shutil.rmtree(work, ignore_errors=True)


def seam_variants(dev, batches, steps, bins, method):
    """The plugin seams INSIDE the unmodified reference, timed on this box: the reference's own BaseGraph +
    TensorRT quantizer + TorchExecutor are the host (driven through oracle/reference_import.py, which only imports and calls
    the staged reference); what runs underneath, and what is measured, is this library:
      kernels : reference pass + reference observers, kernels = libppq_hip.so              (ppq_amd.install_into_ppq)
      pass    : THIS package's RuntimeCalibrationPass + observers in the reference's ppq.lib.Pipeline on the reference's
                executor                                                                   (ppq_amd.install_plugins_into_ppq)
    One warm pass (MIOpen, allocator) + two timed ones each; the better one is reported."""
    stage = staged_reference()
    if stage is None: return []
    out = []
    try:
        import ppq_amd
        from ppq_amd import harness
        from ppq_amd.calibration import RuntimeCalibrationPass as OurPass
        from oracle import reference_import as RI
        RI.load(stage)
        import ppq.lib as PFL
        from ppq.quantization.optim import RuntimeCalibrationPass as RefPass
        n = max(8, steps)                                   # the reference asserts calib_steps >= 8 and cycles the loader
        for stack in ('kernels', 'fast', 'observers', 'pass'):
            ppq_amd.uninstall_from_ppq()
            if stack == 'fast': ppq_amd.install_into_ppq(fast_observers=True)
            elif stack == 'observers': ppq_amd.install_plugins_into_ppq(observers=True)
            else: ppq_amd.install_plugins_into_ppq(observers=False)
            times, checksum = [], None
            for rep in range(3):
                rg, rex = RI.quantize_reference_graph(RI.to_reference_graph(harness.resnet50_graph(seed=0)), dev, batches[0], bins=bins, method=method)
                # small batches: 'auto' -- the pass times one eager step and captures the REFERENCE executor's loop into a HIP graph
                # when that step is launch-bound (batch 1); at batch >= 16 the step is GPU-bound and stays eager, as in the headline
                p = RefPass(method=method) if stack != 'pass' else OurPass(method=method, check_steps=False,
                                                                          use_hip_graph='auto' if batches[0].shape[0] < 16 else False)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                if stack != 'pass': p.optimize(graph=rg, dataloader=batches, executor=rex, calib_steps=n, collate_fn=None)
                else: PFL.Pipeline([p]).optimize(graph=rg, dataloader=batches, executor=rex, calib_steps=n, collate_fn=None, verbose=False)
                torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
                checksum = float(sum(RI.activation_scales(rg).values()))
                replays = getattr(p, 'graph_replays', 0)
                del rg, rex, p
            best = min(times[1:])
            samples = n * batches[0].shape[0]
            out.append({'workload': {'kernels': "UNMODIFIED reference (its executor, RuntimeCalibrationPass and observers) with libppq_hip.so as its kernel extension: install_into_ppq()",
                                     'fast': "the same with the reference's min/max collection wrapped (one ppqhip_minmax_t pass instead of value.min() + value.max()): install_into_ppq(fast_observers=True)",
                                     'observers': "reference executor + the reference's OWN RuntimeCalibrationPass building ppq_amd's observers from its OBSERVER_TABLE: install_plugins_into_ppq(observers=True)",
                                     'pass': "reference executor + BaseGraph driving ppq_amd's RuntimeCalibrationPass + observers inside ppq.lib.Pipeline: install_plugins_into_ppq()"}[stack]
                                    + f'; ResNet-50 {method} {bins} bins, {n} x {batches[0].shape[0]}',
                        'seam': stack, 'samples': samples, 'value': round(samples / best, 2), 'unit': 'samples/s',
                        'ms_per_step': round(best / n * 1e3, 3), 'scale_checksum': checksum, 'graph_replays': replays,
                        'values': [round(samples / t, 2) for t in times[1:]]})
    except Exception as e:            # a broken stage must not cost the bench line
        print(f'[bench] seam variants skipped: {type(e).__name__}: {e}', file=sys.stderr)
    finally:
        try:
            import ppq_amd
            ppq_amd.uninstall_from_ppq()          # the cpu_baseline leg times the reference on ITS OWN torch-CPU path
        except Exception: pass
    return out


def inprocess_variants(dev, args):
    """The headline topology with other calibration settings, timed IN THIS PROCESS (the convolutions are the headline's own:
    MIOpen's find results are already here, no child, no second search): the reference's DEFAULT KL bin count (core/common.py:18:
    4096; BASELINE quotes 2048) and the percentile observer (the quantile launch sequence in situ: one sequence per forward over
    72 tensors, hints from the previous batch).  One warm pass, three timed passes of exactly `steps` steps (median), one more
    pass with hipEvent pairs for the roofline entry of the variant's dominant kernel."""
    from ppq_amd import _lib
    out = []
    for name, bins, method, steps, prefer in (('resnet50_kl_bins4096', 4096, 'kl', 8, None), ('resnet50_percentile', args.bins, 'percentile', 16, ('quantile_t',))):
        try:
            g = torch.Generator(device=dev).manual_seed(777)
            bs = [torch.rand(args.batch, 3, 224, 224, device=dev, generator=g) for _ in range(steps)]
            fresh = lambda: build_workload(dev, bins, method, args.cache_params, args.fuse_params, bool(args.channels_last))      # noqa: E731
            graph, ex = fresh()
            run_pass(graph, ex, bs, steps, method, False, False, True)     # a full-length warm pass: the allocator sees this variant's buffer sizes
            times = []
            for _ in range(5):          # (five: on shared hosts one pass in three or four runs 1.5-2x slow -- r06 run 2: 2904, 1657, 1660 samples/s)
                del graph, ex
                gc.collect()            # the pass objects hold reference cycles: uncollected, the next pass's 600 MB of histogram rows are NEW device allocations
                graph, ex = fresh()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                run_pass(graph, ex, bs, steps, method, False, False, True)
                torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
            del graph, ex
            graph, ex = fresh()
            torch.cuda.synchronize(); _lib.lib.ppqhip_prof_enable(1)
            run_pass(graph, ex, bs, steps, method, False, False, True)
            torch.cuda.synchronize(); _lib.lib.ppqhip_prof_enable(0)
            roof = roofline_entry(collect_prof(), prefer=prefer) or {}
            del graph, ex, bs
            med = sorted(times)[len(times) // 2]
            n = steps * args.batch
            out.append({'workload': f'ResNet-50 topology, RuntimeCalibrationPass {method} {bins} bins, {steps} batches x {args.batch} (in process)',
                        'name': name, 'value': round(n / med, 2), 'unit': 'samples/s', 'ms_per_step': round(med / steps * 1e3, 3), 'steps': steps,
                        'values': [round(n / t, 2) for t in times], 'spread_pct': round(100.0 * (max(times) - min(times)) / med, 2),
                        'roofline': {k: roof.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'launches', 'avg_launch_us',
                                                              'algorithmic_bytes_per_launch')}})
        except Exception as e:
            out.append({'workload': name, 'name': name, 'error': f'{type(e).__name__}: {e}'})
        torch.cuda.empty_cache()
    return out


def workload_variants(args):
    """BASELINE configs 3, 4, 5 as short child runs of this file: each line's value + the roofline entry of its dominant
    kernel.  Children skip baselines, PMC passes and their own variants."""
    out = []
    for name, extra in (('resnet50_cfg3', ['--workload', 'resnet50_cfg3', '--steps', '8', '--batch', '32']),
                        ('vit_b16_fp8', ['--workload', 'vit_b16_fp8', '--steps', '8', '--batch', '16']),
                        # 64 optimizer steps per block since round 6 (rounds 4-5: 8; the reference's default: 500, optim/training.py:700-713).
                        # At 8 steps the 27 captures + eager first steps are 2/3 of the 0.24 s pass and its run-to-run spread was
                        # 10-20 %; at 64 the replays are the pass: seven passes spread 0.4 % (profiles/r06_cfg5_steps.txt).  The
                        # figure is NOT comparable with the 8-step one of earlier rounds (7100-7700 samples/s there, 18100 here).
                        ('yolov6s_int4_lsq', ['--workload', 'yolov6s_int4_lsq', '--steps', '64', '--batch', '8']),
                        ):
        # every child warms ITSELF with MIOpen's find mode on (one complete untimed pass over one batch): the parent's own find
        # results live in its process until it exits, and a child in immediate mode on a fresh box runs the vendor library's
        # heuristic picks instead (BENCH_r04: config 3 at 26 ms / step in the driver's run against 11.5 in the builder's, whose
        # box had a user database from an earlier command); >= 8 timed steps, three timed passes (value = their median), spread reported
        # (config 5 stays in immediate mode: the search over 56 convolutions x forward / data-gradient / weight-gradient costs the
        #  child 160 s -- 183 s against 22 s of wall time -- and buys 6 % per replayed step: 0.330 vs 0.349 ms)
        find = '0' if name == 'yolov6s_int4_lsq' else '1'
        # config 5's timed pass is 0.24 s of host-driven work (27 captures, 27 eager first steps, 189 replays): three passes spread 10-20 %
        # by max - min (BENCH_r05: 21.9 %); seven passes, value = their median, spread = (p90 - p10) / median, all seven in `values`
        reps = '7' if name == 'yolov6s_int4_lsq' else '3'
        cmd = [sys.executable, os.path.abspath(__file__), '--warmup', '1', '--repeats', reps, '--variants', '0', '--pmc', '0',
               '--no-cpu-baseline', '--no-cpu-ops', '--settle-ms', '100', '--miopen-find', find] + extra
        try:
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            if r.returncode != 0 or not line:
                out.append({'workload': name, 'error': (r.stderr or r.stdout)[-300:]}); continue
            j = json.loads(line[-1])
            roof = j.get('roofline') or {}
            out.append({'workload': j['config']['workload'], 'name': name, 'metric': j['metric'], 'value': j['value'], 'unit': j['unit'],
                        'ms_per_step': j['ms_per_step'], 'steps': j['steps'], 'values': j.get('values'), 'spread_pct': j.get('spread_pct'),
                        'range_pct': j.get('range_pct'),
                        'child_wall_s': round(time.perf_counter() - t0, 1),
                        'roofline': {k: roof.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'launches', 'avg_launch_us',
                                                              'algorithmic_bytes_per_launch')}})
            if name == 'yolov6s_int4_lsq':      # per-block tensors of 0.05 .. 6 MB: every launch is latency-, not bandwidth-bound
                out[-1]['execution'] = j['config'].get('execution')
                out[-1]['launches_per_eager_step'] = j['config'].get('launches_per_eager_step')
                out[-1]['per_tensor_eager_samples_per_s'] = j['config'].get('per_tensor_eager_samples_per_s')
                out[-1]['phase_ms_whole_pass'] = j['config'].get('phase_ms_whole_pass')
                out[-1]['step'] = j['config'].get('step')
                out[-1]['roofline']['note'] = ('block-wise finetuning: all weight delegators of a block share ONE LSQ-backward launch per step '
                                               '(ppqhip_fq_linear_c_bwd_multi; activations keep one launch each) and the step is replayed from a '
                                               'HIP graph; per-block tensors of 0.05 .. 6 MB stay latency-bound; the same kernels on '
                                               '[32,512,56,56]: 0.69-0.72 of 8 TB/s (profiles/r03_microbench_randn.txt)')
        except Exception as e:
            out.append({'workload': name, 'error': f'{type(e).__name__}: {e}'})
    return out


def main_lsq(args, rank, world, dev):
    """--workload yolov6s_int4_lsq (BASELINE config 5): a *step* is one optimizer step of the block-wise LearnedStepSizePass
    on one batch (forward through the block with the HIP fake-quant kernels, backward through the LSQ kernels, Adam); the
    timed region is one LearnedStepSizePass.optimize over all 27 blocks with `--steps` steps each (it includes the pass's own
    collection of FP32 targets / quantised block inputs).  value = blocks x steps x batch / time, aggregated over ranks
    (data parallel: each rank its own batches, ONE flat gradient all-reduce per step)."""
    import ppq_amd  # noqa: F401
    from ppq_amd import _lib, harness
    from ppq_amd.calibration import RuntimeCalibrationPass
    from ppq_amd.lsq import LearnedStepSizePass
    size = 320
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    batches = [torch.rand(args.batch, 3, size, size, device=dev, generator=g) for _ in range(4)]
    group = None
    if world > 1:
        import torch.distributed as dist
        group = dist.group.WORLD

    def build():
        graph = harness.yolov6s_graph(seed=3)
        harness.quantize_graph(graph, 'minmax')
        for op in graph.operations.values():                        # weights -> int4 [-8, 7]
            for cfg, var in op.config_with_variable:
                if var.is_parameter and cfg.state.value == 1: cfg.num_of_bits, cfg.quant_min, cfg.quant_max = 4, -8, 7
        ex = harness.TorchExecutor(graph, dev)
        harness.ParameterQuantizePass().optimize(graph)
        RuntimeCalibrationPass(check_steps=False).optimize(graph, dataloader=batches, executor=ex, calib_steps=len(batches))
        return graph, ex
    if args.warmup > 0:
        graph, ex = build()
        LearnedStepSizePass(steps=1, lr=1e-5, block_size=5, process_group=group).optimize(graph, batches[:1], ex)
        del graph, ex
    if args.kt_child:            # under rocprofv3 --kernel-trace (count_dispatches): ONE pass of `--steps` steps per block, nothing else
        graph, ex = build()
        LearnedStepSizePass(steps=args.steps, lr=1e-5, block_size=5).optimize(graph, batches, ex)
        torch.cuda.synchronize()
        return
    times, p = [], None
    for rep in range(max(1, args.repeats)):
        graph, ex = build()
        p = LearnedStepSizePass(steps=args.steps, lr=1e-5, block_size=5, process_group=group)
        barrier(world); t0 = time.perf_counter()
        p.optimize(graph, batches, ex)
        barrier(world); elapsed = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())
        times.append(elapsed)
        del ex
    elapsed = sorted(times)[len(times) // 2]
    blocks = len(p.report)
    roof, prof_rows, launches_per_step, per_tensor_eager_s, phase_ms, step_report = None, [], None, 1.0, None, None
    if rank == 0 and world == 1:
        graph2, ex2 = build()
        torch.cuda.synchronize(); _lib.lib.ppqhip_prof_enable(1)
        # launch statistics come from an EAGER pass (events cannot time launches that sit inside a replayed HIP graph)
        pp = LearnedStepSizePass(steps=min(args.steps, 4), lr=1e-5, block_size=5, use_hip_graph=False)
        pp.optimize(graph2, batches, ex2)
        torch.cuda.synchronize(); _lib.lib.ppqhip_prof_enable(0)
        prof_rows = collect_prof()
        roof = roofline_entry(prof_rows, prefer=('fq_linear_c_bwd', 'fq_linear_t_bwd'))
        eager_steps = max(1, pp.stats['eager_steps'])
        launches_per_step = {r['name']: round(r['launches'] / eager_steps, 2) for r in prof_rows}
        # where the pass spends its time (synchronised phases; a separate pass, not the timed one)
        graph4, ex4 = build()
        pq = LearnedStepSizePass(steps=args.steps, lr=1e-5, block_size=5)
        pq.profile_phases = True
        pq.optimize(graph4, batches, ex4)
        phase_ms = {k: round(v, 1) for k, v in pq.phase_ms.items()}
        del graph4, ex4

        def replay_ms(steps, max_blocks, **kw):
            """ms per REPLAYED optimizer step (synchronised around the replay loops of all blocks)."""
            gq, eq = build()
            q = LearnedStepSizePass(steps=steps, lr=1e-5, block_size=5, **kw)
            q.profile_phases, q.max_blocks = True, max_blocks
            q.optimize(gq, batches, eq)
            n = max(1, q.stats['graph_replays'])
            return round(q.phase_ms.get('graph_replays', 0.0) / n, 4), q.stats
        # what the round-5 changes to the step buy: activation delegators with ONE finish launch per step + fused Adam, against
        # round 4's step (a finish launch per activation, foreach Adam) -- same blocks, same steps
        r5_ms, r5_stats = replay_ms(args.steps, None)
        r4_ms, _ = replay_ms(args.steps, None, group_activations=False, fused_adam=False)
        # the reference's DEFAULT step count (500, optim/training.py:700-713) on the first 3 blocks: the regime in which the
        # replays, not the collection / capture overheads, are the pass
        torch.cuda.synchronize(); t5 = time.perf_counter()
        long_ms, long_stats = replay_ms(500, 3)
        torch.cuda.synchronize(); t5 = time.perf_counter() - t5
        step_report = {'replay_ms_per_step': r5_ms, 'replay_ms_per_step_round4_form': r4_ms,
                       'fused_adam_blocks': r5_stats.get('fused_adam_blocks', 0), 'grouped_activations': r5_stats.get('grouped_activations', 0),
                       'steps500_blocks3': {'replay_ms_per_step': long_ms, 'wall_s': round(t5, 2),
                                            'samples_per_s': round(3 * 500 * args.batch / t5, 1), 'graph_replays': long_stats['graph_replays']}}
        # kernel dispatches of ONE replayed step: two kernel-traced child passes that differ only in the steps per block (8 and 24);
        # everything else -- calibration, collection, the eager first step and the capture of each block -- cancels in the difference
        # (MIOpen in immediate mode, as the driver's line runs this child: the count includes whatever helper kernels the picked
        #  convolution solvers launch, and a find-mode search over 56 convolutions x 3 directions under the tracer ran into the 240 s limit)
        kd_args = ['--workload', 'yolov6s_int4_lsq', '--batch', str(args.batch), '--warmup', '1', '--kt-child', '--variants', '0', '--pmc', '0',
                   '--no-cpu-baseline', '--no-cpu-ops', '--miopen-find', '0']
        n8, note8 = count_dispatches(kd_args + ['--steps', '8'])
        n24, note24 = count_dispatches(kd_args + ['--steps', '24'])
        if n8 is not None and n24 is not None:
            step_report['launches_per_replayed_step'] = round((n24 - n8) / (16.0 * blocks), 2)
            step_report['launches_per_replayed_step_source'] = (f'rocprofv3 --kernel-trace dispatch counts of two child passes: {n24} at 24 steps per block, '
                                                                f'{n8} at 8, over {blocks} blocks')
        else:
            step_report['launches_per_replayed_step'] = None
            step_report['launches_per_replayed_step_source'] = f'{note8}; {note24}'
        # the same pass without this round's execution choices: per-tensor launches, eager steps (what round 3 measured)
        graph3, ex3 = build()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        pr = LearnedStepSizePass(steps=args.steps, lr=1e-5, block_size=5, group_weights=False, use_hip_graph=False)
        pr.incremental_inputs = False
        pr.optimize(graph3, batches, ex3)
        torch.cuda.synchronize(); per_tensor_eager_s = time.perf_counter() - t1
    if rank == 0:
        total_steps = blocks * args.steps
        samples = world * total_steps * args.batch
        print(json.dumps({
            'metric': 'LSQ finetune samples/sec (block-wise LearnedStepSizePass, YOLOv6-s-like INT4 weights)', 'value': round(samples / elapsed, 2),
            'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / total_steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 (INT4 / INT8 simulated)', 'data': 'synthetic',
            'repeats': len(times), 'values': [round(samples / t, 2) for t in times],
            # >= 7 passes: the spread is the inter-decile range (the second smallest / second largest of seven), the full range is kept too
            'spread_pct': round(100.0 * ((sorted(times)[-2] - sorted(times)[1]) if len(times) >= 7 else (max(times) - min(times))) / elapsed, 2),
            'range_pct': round(100.0 * (max(times) - min(times)) / elapsed, 2),
            'config': {'workload': f'{WORKLOADS[WORKLOAD][0]}; {blocks} blocks x {args.steps} Adam steps x batch {args.batch} x 3x{size}x{size} per GPU '
                                   f'(timed: the whole pass incl. its target / input collection)', 'samples': samples, 'batch': args.batch,
                       'blocks': blocks, 'optimizer_steps': total_steps, 'kept_blocks': sum(1 for _, a, b in p.report if b <= a),
                       'parallelism': f'dp{world} (one flat gradient all-reduce per step)', 'rccl_ranks': world,
                       'execution': dict(p.stats, graph_error=p.graph_error, note='grouped_weights: weight delegators served by ONE forward + ONE backward launch per step; '
                                                        'graph_replays: optimizer steps replayed from a captured HIP graph'),
                       'launches_per_eager_step': launches_per_step if (rank == 0 and world == 1) else None,
                       'phase_ms_whole_pass': phase_ms, 'step': step_report,
                       'replay_ms_per_step': (step_report or {}).get('replay_ms_per_step'),
                       'replay_ms_per_step_round4_form': (step_report or {}).get('replay_ms_per_step_round4_form'),
                       'steps500_blocks3_samples_per_s': ((step_report or {}).get('steps500_blocks3') or {}).get('samples_per_s'),
                       'per_tensor_eager_samples_per_s': round(samples / per_tensor_eager_s, 2) if (rank == 0 and world == 1) else None},
            'roofline': roof, 'cpu_baseline': None,
            'kernels': [{'name': r['name'], 'launches': r['launches'], 'total_ms': round(r['total_ms'], 3),
                         'GBps': round(r['total_bytes'] / max(r['total_ms'], 1e-9) / 1e6, 1)} for r in prof_rows]}), file=JSON_OUT, flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--total-samples', type=int, default=0,
                    help='STRONG scaling: a fixed job of this many samples sharded over the ranks (steps = total / (ranks x batch), overrides --steps; '
                         'BASELINE config 3 = --workload resnet50_cfg3 --total-samples 1024 --gpus 8); scaling is reported as "strong"')
    ap.add_argument('--bins', type=int, default=2048)
    ap.add_argument('--method', type=str, default='kl')
    ap.add_argument('--workload', type=str, default='resnet50', choices=sorted(WORKLOADS),
                    help='resnet50 = BASELINE config 2 (the metric); resnet50_cfg3 / vit_b16_fp8 = configs 3 / 4 at size')
    ap.add_argument('--repeats', type=int, default=5, help='timed passes (each exactly K steps); value = median')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cpu-ops', action='store_true')
    ap.add_argument('--variants', type=int, default=1, help="also time SURVEY 8(d)'s own protocol (batch 1 x 256 steps) once and report it in config.variants")
    ap.add_argument('--pmc', type=int, default=1, help='measure roofline.traffic with two rocprofv3 --pmc child passes of this command')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--kt-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--host-selftest', action='store_true', help='CPU-only check of spawn + rendezvous + merge (gloo); no GPU work')
    ap.add_argument('--backend', type=str, default='nccl')
    ap.add_argument('--single-device', type=int, default=0, help='debug: put every rank on cuda:0 (use with --backend gloo)')
    ap.add_argument('--hip-graph', default='auto', choices=['0', '1', 'auto'],
                    help="replay each phase's forward as a HIP graph: never / always / auto = only for small batches "
                         "(< 16 samples) whose timed eager step is launch-bound")
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

    global WORKLOAD
    WORKLOAD = args.workload
    if args.total_samples > 0:          # strong scaling: the job is fixed, every rank takes total / (ranks x batch) batches of it
        ranks = int(os.environ.get('WORLD_SIZE', max(1, args.gpus)))
        per = ranks * args.batch
        if args.total_samples % per: raise SystemExit(f'--total-samples {args.total_samples} is not a multiple of ranks x batch = {ranks} x {args.batch}')
        args.steps = args.total_samples // per
    per_rank_miopen_dirs()
    if WORKLOADS[WORKLOAD][1] is not None: args.method = WORKLOADS[WORKLOAD][1]
    if WORKLOAD != 'resnet50':              # the reference-CPU legs and the batch-1 variant describe config 2 only
        args.no_cpu_baseline = args.no_cpu_ops = True
        args.variants = 0
    maybe_spawn(args)
    if args.host_selftest: return host_selftest(args)
    rank, world, local = setup_dist(args.gpus, args.backend, bool(args.single_device))
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    dev = f'cuda:{local}'
    import ppq_amd  # noqa: F401  (fails loudly without libppq_hip.so)
    from ppq_amd import _lib
    if WORKLOAD == 'yolov6s_int4_lsq': return main_lsq(args, rank, world, dev)

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    batches = [torch.rand(args.batch, 3, 224, 224, device=dev, generator=g) for _ in range(args.steps)]
    # HIP-graph replay only pays for launch-bound steps (small batches); at batch >= 16 the step is GPU-bound
    # and 'auto' used to flip on a noisy host-enqueue measurement (round 1's driver run: 15.0 vs 11.0 ms/step)
    hip_graph = {'0': False, '1': True, 'auto': 'auto' if args.batch < 16 else False}[args.hip_graph]

    # warm-up: W batches through a complete two-phase pass (MIOpen find, library load, allocator)
    if args.warmup > 0:
        graph, ex = build_workload(dev, args.bins, args.method, args.cache_params, args.fuse_params, bool(args.channels_last))
        run_pass(graph, ex, batches[: max(1, min(args.warmup, args.steps))], max(1, min(args.warmup, args.steps)), args.method,
                 bool(args.async_observe), False, bool(args.batch_observations))
        # keep the device under the workload's own load profile for a moment: a GPU that sat idle (fresh
        # box) otherwise spends the first timed steps in clock / power transitions
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for b in batches[:2]: ex.forward(b)
            torch.cuda.synchronize()
        del graph, ex

    # timed region(s): `repeats` independent passes of exactly K steps each
    global TRACE_STEPS
    times, trace, p, graph = [], None, None, None
    host_passes = []          # per timed pass: how long the container was parked by its CPU quota, how many CPUs it kept busy
    repeats = 1 if (args.trace_steps or args.pmc_child or args.kt_child) else max(1, args.repeats)
    for rep in range(repeats):
        if args.trace_steps:
            TRACE_STEPS = []
            ev0 = torch.cuda.Event(enable_timing=True)
        del p, graph
        graph, ex = build_workload(dev, args.bins, args.method, args.cache_params, args.fuse_params, bool(args.channels_last))
        barrier(world)
        cg0 = cgroup_cpu_stat()
        t0 = time.perf_counter()
        if args.trace_steps: ev0.record()
        p = run_pass(graph, ex, batches, args.steps, args.method, bool(args.async_observe), hip_graph,
                     bool(args.batch_observations), bool(args.reuse_activations), (args.queue_mib << 20) or None)
        barrier(world)
        elapsed = time.perf_counter() - t0
        cg1 = cgroup_cpu_stat()
        if cg0 and cg1:
            host_passes.append({'throttled_ms': round((cg1.get('throttled_usec', 0) - cg0.get('throttled_usec', 0)) / 1e3, 1),
                                'cpus_busy': round((cg1.get('usage_usec', 0) - cg0.get('usage_usec', 0)) / 1e6 / max(elapsed, 1e-9), 1)})
        if args.trace_steps:
            trace = [{'host_ms': round((t - t0) * 1e3, 2), 'dev_ms': round(ev0.elapsed_time(e), 2)} if not isinstance(e, float)
                     else {'render_start_ms': round((t - t0) * 1e3, 2), 'render_ms': round((e - t) * 1e3, 2)} for t, e in TRACE_STEPS]
            TRACE_STEPS = None
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        times.append(elapsed)
        del ex
    elapsed = sorted(times)[len(times) // 2]
    if args.pmc_child or args.kt_child: return     # the rocprofv3 children: counters / kernel trace only, no JSON line
    n_obs = sum(1 for op in graph.operations.values() if hasattr(op, 'config') for c, v in op.config_with_variable
                if not v.is_parameter and int(getattr(c.state, 'value', c.state)) == 4)      # activation configs this pass calibrated
    scale_checksum = float(sum(float(c.scale.sum()) for op in graph.operations.values() if hasattr(op, 'config')
                               for c, v in op.config_with_variable if not v.is_parameter and c.scale is not None
                               and int(getattr(c.state, 'value', c.state)) == 4))
    merge_stats = [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in m.items()} for m in p.merge_stats]

    # roofline leg: the identical pass once more with hipEvent pairs around every library launch
    # (measured right behind the timed passes, before the variants churn the allocator: the same tensors in the same memory state)
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
        del graph2, ex2
        roof = roofline_entry(prof_rows, prefer={'vit_b16_fp8': ('fq_float',), 'resnet50_cfg3': ('hist_asym_t',)}.get(WORKLOAD))

    # BASELINE config 2 exactly as SURVEY 8(d) words it: batch 1, calib_steps 256 (launch-bound: HIP-graph replay pays)
    variants = []
    if args.variants and world == 1 and not (args.batch == 1 and args.steps == 256):
        vb = [torch.rand(1, 3, 224, 224, device=dev, generator=g) for _ in range(256)]
        vtimes = []
        for _ in range(2):                 # first pass warms MIOpen's batch-1 kernels, second is reported
            graph_v, ex_v = build_workload(dev, args.bins, args.method, args.cache_params, args.fuse_params, bool(args.channels_last))
            barrier(world); tv = time.perf_counter()
            parts = {}
            pv = run_pass(graph_v, ex_v, vb, 256, args.method, False, 'auto', bool(args.batch_observations), decompose=parts)
            barrier(world); vtimes.append(time.perf_counter() - tv)
            del graph_v, ex_v
        # where the wall time goes: the two forward loops (256 forwards each; the first steps eager, the rest one HIP-graph
        # replay per step when 'auto' found the eager step launch-bound: issue_ms = host time to enqueue one eager forward,
        # total_ms = until the device drained it) and the two renders (range fetch; histogram fold + batched KL search)
        variants.append({'workload': 'batch 1 x 256 steps (SURVEY 8(d) protocol), HIP-graph replay auto', 'samples': 256,
                         'value': round(256 / vtimes[-1], 2), 'unit': 'samples/s', 'ms_per_step': round(vtimes[-1] / 256 * 1e3, 3),
                         'graph_replays': pv.graph_replays,
                         'decomposition': {'wall_ms': round(vtimes[-1] * 1e3, 1), **parts,
                                           'ms_per_forward': [round(v / 256, 3) for v in parts.get('forward_loops_ms', [])],
                                           'eager_step': [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items()}
                                                          for d in pv.graph_decisions]}})
        del vb, pv
    if args.variants and world == 1 and not args.reuse_activations:
        # MI355X-first variant of the SAME pass (not the headline: the reference's protocol runs every batch twice):
        # phase 2 bins the phase-1 activations kept resident in HBM instead of recomputing them with a second forward
        # one untimed pass first: the 41 GB of kept activations are NEW allocations (hipMalloc, not the caching allocator's
        # free list) the first time round -- 0.8 s of driver time that says nothing about the pass (r4: 682 vs 5595 samples/s)
        graph_w, ex_w = build_workload(dev, args.bins, args.method, args.cache_params, args.fuse_params, bool(args.channels_last))
        run_pass(graph_w, ex_w, batches, args.steps, args.method, False, False, True, True)
        del graph_w, ex_w
        graph_v, ex_v = build_workload(dev, args.bins, args.method, args.cache_params, args.fuse_params, bool(args.channels_last))
        barrier(world); tv = time.perf_counter()
        pv = run_pass(graph_v, ex_v, batches, args.steps, args.method, False, False, True, True)
        barrier(world); tv = time.perf_counter() - tv
        check_v = float(sum(float(c.scale.sum()) for op in graph_v.operations.values() if hasattr(op, 'config')
                            for c, v in op.config_with_variable if not v.is_parameter and c.scale is not None
                            and int(getattr(c.state, 'value', c.state)) == 4))
        variants.append({'workload': f'the same {args.steps} x {args.batch} pass with reuse_activations: phase 2 bins the phase-1 activations '
                                     'kept in HBM, no second forward (scale checksum equal to ~1e-7 relative: the two forwards of the headline are not bitwise repeatable, MIOpen atomics)', 'samples': args.steps * args.batch,
                         'value': round(args.steps * args.batch / tv, 2), 'unit': 'samples/s', 'ms_per_step': round(tv / args.steps * 1e3, 3),
                         'replayed_batches': pv.replayed_batches, 'resident_MiB': round(pv.replay_peak_bytes / 2 ** 20),
                         'scale_checksum': check_v})
        del graph_v, ex_v, pv

    if args.variants and world == 1 and WORKLOAD == 'resnet50':
        variants += seam_variants(dev, batches, args.steps, args.bins, args.method)
        # the same two seams at SURVEY 8(d)'s literal protocol size: batch 1 (64 of the 256 samples: host time per tensor call
        # is the step here -- the facade's share of it: tools/call_overhead.py, profiles/r04_call_overhead.txt)
        g1 = torch.Generator(device=dev).manual_seed(99)
        variants += seam_variants(dev, [torch.rand(1, 3, 224, 224, device=dev, generator=g1) for _ in range(64)], 64, args.bins, args.method)
        torch.cuda.empty_cache()
        variants += inprocess_variants(dev, args)
        variants += workload_variants(args)

    if world > 1:
        barrier(world)
    torch.cuda.empty_cache()
    if roof is not None and world == 1 and args.pmc:
        child = ['--pmc-child', '--no-cpu-baseline', '--no-cpu-ops', '--pmc', '0', '--variants', '0', '--steps', str(min(args.steps, 2)),
                 '--warmup', '0', '--repeats', '1', '--batch', str(args.batch), '--bins', str(args.bins),
                 '--method', args.method, '--workload', args.workload, '--hip-graph', '0', '--miopen-find', '0', '--fuse-params', str(args.fuse_params),
                 '--batch-observations', str(args.batch_observations), '--channels-last', str(args.channels_last)]
        roof['traffic'], roof['traffic_source'] = pmc_traffic(roof['kernel'], child)
        kt_child = [a for a in child if a != '--pmc-child']
        kt_child[kt_child.index('--steps') + 1] = str(min(args.steps, 4))
        med, n_med, note = rocprof_kernel_median(roof['kernel'], kt_child + ['--kt-child'])
        roof['rocprof_median_us'], roof['rocprof_launches'], roof['rocprof_source'] = med, n_med, note
        if med:
            roof['rocprof_frac'] = round(roof['algorithmic_bytes_per_launch'] / (med * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)

    cpu = cpu_ops = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.bins)
    if rank == 0 and world == 1 and not args.no_cpu_ops:
        from oracle import torch_cpu_path as T
        cpu_ops = {'kind': 'reference-path', 'cores': torch.get_num_threads(), 'unit': 'ms (median) / GB/s of algorithmic bytes',
                   'rows': T.op_table(args.bins)}

    scalars = {}
    if rank == 0 and world == 1 and args.variants:
        # the driver keeps the SCALAR keys of `config` and drops nested ones: the figures that matter are repeated here flat
        def val(pred):
            for v in variants:
                if pred(v) and 'value' in v: return v['value']
            return None
        scalars = {
            'batch1_x256_samples_per_s': val(lambda v: str(v.get('workload', '')).startswith('batch 1 x 256')),
            'reuse_activations_samples_per_s': val(lambda v: 'reuse_activations' in str(v.get('workload', ''))),
            f'seam_kernels_b{args.batch}_samples_per_s': val(lambda v: v.get('seam') == 'kernels' and v.get('samples', 0) > 64),
            f'seam_pass_b{args.batch}_samples_per_s': val(lambda v: v.get('seam') == 'pass' and v.get('samples', 0) > 64),
            f'seam_fast_b{args.batch}_samples_per_s': val(lambda v: v.get('seam') == 'fast' and v.get('samples', 0) > 64),
            f'seam_observers_b{args.batch}_samples_per_s': val(lambda v: v.get('seam') == 'observers' and v.get('samples', 0) > 64),
            'seam_kernels_b1_samples_per_s': val(lambda v: v.get('seam') == 'kernels' and v.get('samples', 0) == 64),
            'seam_pass_b1_samples_per_s': val(lambda v: v.get('seam') == 'pass' and v.get('samples', 0) == 64),
            'cfg3_samples_per_s': val(lambda v: v.get('name') == 'resnet50_cfg3'),
            'cfg4_samples_per_s': val(lambda v: v.get('name') == 'vit_b16_fp8'),
            'cfg5_samples_per_s': val(lambda v: v.get('name') == 'yolov6s_int4_lsq'),
            'cfg5_steps_per_block': next((v.get('steps') for v in variants if v.get('name') == 'yolov6s_int4_lsq'), None),
            'cfg5_replay_ms_per_step': next((((v.get('step') or {}).get('replay_ms_per_step')) for v in variants if v.get('name') == 'yolov6s_int4_lsq'), None),
            'cfg5_launches_per_replayed_step': next((((v.get('step') or {}).get('launches_per_replayed_step')) for v in variants if v.get('name') == 'yolov6s_int4_lsq'), None),
            'cfg5_replay_ms_per_step_round4_form': next((((v.get('step') or {}).get('replay_ms_per_step_round4_form')) for v in variants if v.get('name') == 'yolov6s_int4_lsq'), None),
            'cfg5_steps500_blocks3_samples_per_s': next(((((v.get('step') or {}).get('steps500_blocks3') or {}).get('samples_per_s')) for v in variants if v.get('name') == 'yolov6s_int4_lsq'), None),
            'kl4096_samples_per_s': val(lambda v: v.get('name') == 'resnet50_kl_bins4096'),
            'percentile_samples_per_s': val(lambda v: v.get('name') == 'resnet50_percentile'),
        }
        for v in variants:
            if v.get('name') in ('resnet50_cfg3', 'vit_b16_fp8', 'yolov6s_int4_lsq') and v.get('roofline'):
                key = {'resnet50_cfg3': 'cfg3', 'vit_b16_fp8': 'cfg4', 'yolov6s_int4_lsq': 'cfg5'}[v['name']]
                scalars[f'{key}_roofline_frac'] = v['roofline'].get('frac')
                scalars[f'{key}_spread_pct'] = v.get('spread_pct')
                if v.get('range_pct') is not None: scalars[f'{key}_range_pct'] = v.get('range_pct')
        scalars.update(north_star_b(args.bins))   # B_* keys: rocprofv3 medians + graph-replay bound + floors + B_status
        # config 5's dominant launch against the floor OF ITS SIZE rather than against 8 TB/s alone: its launches move ~10 MB, where a
        # launch is a latency chain -- the floor for that many bytes is read off this box's own floors (the empty kernel + the share of
        # B's copy floor, 12.85 MB of traffic, that the launch's bytes are)
        for v in variants:
            roofv = v.get('roofline') or {}
            # (the profiler-free floors -- HIP-graph replays -- beside the event-timed launch: the tracer's own medians sit 0.3-0.5 us higher)
            if v.get('name') == 'yolov6s_int4_lsq' and roofv.get('avg_launch_us') and scalars.get('B_floor_copy_graph_us') and scalars.get('B_floor_empty_graph_us'):
                e, c = scalars['B_floor_empty_graph_us'], scalars['B_floor_copy_graph_us']
                floor_us = e + (c - e) * roofv['algorithmic_bytes_per_launch'] / (8.0 * 512 * 56 * 56)
                scalars['cfg5_dominant_launch_us'] = roofv['avg_launch_us']
                scalars['cfg5_dominant_launch_floor_us'] = round(floor_us, 2)
                scalars['cfg5_dominant_launch_over_floor'] = round(roofv['avg_launch_us'] / floor_us, 2)
    if rank == 0:
        samples = world * args.steps * args.batch
        out = {
            'metric': ('calibration samples/sec (RuntimeCalibrationPass, KL 2048-bin, ResNet-50 INT8)' if (args.method == 'kl' and args.bins == 2048 and WORKLOAD == 'resnet50')
                       else f'calibration samples/sec (RuntimeCalibrationPass, {args.method}, {args.bins} bins, {WORKLOAD})'),
            'value': round(samples / elapsed, 2), 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'strong' if args.total_samples > 0 else 'weak', 'vs_baseline': None,
            'dtype': 'f32' if WORKLOAD != 'vit_b16_fp8' else 'f32 (FP8 E4M3 simulated)', 'data': 'synthetic',
            'repeats': len(times), 'values': [round(samples / t, 2) for t in times],
            'spread_pct': round(100.0 * (max(times) - min(times)) / elapsed, 2),
            # the host side of the timed passes: CPUs the container may use (affinity cut by its cgroup quota), the thread pools sized
            # to that, and per pass how long the kernel parked the container for exceeding the quota -- a pass with tens of ms here
            # is a host artefact (value = the median pass)
            'host': {'cpus_visible': os.cpu_count(), 'cpus_granted': HOST_CPUS, 'torch_threads': torch.get_num_threads(),
                     'passes': host_passes},
            'config': {'workload': (f'ResNet-50 topology (53 Conv + 1 Gemm, BN folded, seeded He init), '
                                    f'RuntimeCalibrationPass {args.method} {args.bins} bins, per-tensor INT8 activations, '
                                    f'per-channel INT8 weights, {args.steps} batches x {args.batch} x 3x224x224 per GPU'
                                    + (f' = a fixed job of {args.total_samples} samples sharded over {world} ranks' if args.total_samples else '')) if WORKLOAD == 'resnet50'
                       else (f'{WORKLOADS[WORKLOAD][0]}; RuntimeCalibrationPass {args.method}, {args.bins} bins, {args.steps} batches x {args.batch} x 3x224x224 per GPU'
                             + (f' = a fixed job of {args.total_samples} samples sharded over {world} ranks' if args.total_samples else '')),
                       'samples': samples, 'batch': args.batch, 'observed_tensors': n_obs,
                       'parallelism': f'dp{world} (batches sharded; per phase 1 layout-probe MAX + 1 data all-reduce over one flat buffer)',
                       'rccl_ranks': world, 'backend': args.backend if world > 1 else None, 'merge': merge_stats,
                       'total_samples': args.total_samples or None,
                       # the merges as flat scalars (the driver keeps scalar keys): one layout probe + one data all-reduce per phase
                       'merge_phase1_ms': merge_stats[0].get('ms') if len(merge_stats) > 0 else None,
                       'merge_phase2_ms': merge_stats[1].get('ms') if len(merge_stats) > 1 else None,
                       'merge_collectives_per_phase': merge_stats[0].get('collectives') if merge_stats else None,
                       'merge_phase1_bytes': (merge_stats[0].get('min_f32_bytes') if merge_stats else None),
                       'merge_phase2_bytes': (merge_stats[1].get('sum_int32_bytes') if len(merge_stats) > 1 else None),
                       **scalars,
                       'variants': variants,
                       'async_observe': bool(args.async_observe), 'cache_params': bool(args.cache_params),
                       'channels_last': bool(args.channels_last), 'fuse_params': bool(args.fuse_params), 'batch_observations': bool(args.batch_observations),
                       'reuse_activations': bool(args.reuse_activations), 'replayed_batches': p.replayed_batches,
                       'hip_graph': args.hip_graph, 'graph_replays': p.graph_replays,
                       'graph_decisions': [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items()}
                                           for d in p.graph_decisions]},
            'roofline': roof, 'cpu_baseline': cpu, 'cpu_ops': cpu_ops,
            'kernels': [{'name': r['name'], 'launches': r['launches'], 'total_ms': round(r['total_ms'], 3),
                         'GBps': round(r['total_bytes'] / max(r['total_ms'], 1e-9) / 1e6, 1)} for r in prof_rows],
            'scale_checksum': scale_checksum,
        }
        if trace is not None: out['trace_steps'] = trace
        print(json.dumps(out), file=JSON_OUT, flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    # stdout carries ONE line, the JSON; whatever a library prints on the way (the staged reference greets with a banner) goes to stderr
    sys.stdout = sys.stderr
    main()
