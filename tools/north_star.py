"""The single-tensor launches on the north star's tensor B = [1,512,56,56] fp32 (6.4 MB) and on B x {2,4,8,16,32}, each priced
against FLOOR kernels measured the same way in the same process (tools/floor/floor_kernels.hip: an empty kernel, a pure read, a
copy, a read followed by the cross-workgroup atomic combine).  Two measurement methods, same case list:

  trace : this file run under `rocprofv3 --kernel-trace` (`--mode trace-child`); every case is preceded by one launch of
          `floor_marker_kernel` whose grid size is the case id, the parent cuts the trace at the markers and reports the MEDIAN
          begin->end device duration per call (a call that launches several kernels = the sum of their medians).
  graph : profiler-free -- `iters` back-to-back launches of the case captured into ONE HIP graph, replayed, timed with one event
          pair, divided by `iters`.  A launch-to-launch time inside a graph: an UPPER bound of the kernel duration (it includes
          the inter-node gap, which the empty-kernel case prices).  Exists on any box; what bench.py falls back to.

bench.py imports `measure()`; standalone:
  python tools/north_star.py --mode both --sizes 1,2,4,8,16,32 --report profiles/r06_frac_vs_size.txt
"""
import argparse
import collections
import csv
import ctypes
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path: sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0
B_SHAPE = (1, 512, 56, 56)
SIZES = (1, 2, 4, 8, 16, 32)
ITERS = {1: 200, 2: 120, 4: 100, 8: 60, 16: 40, 32: 30}
ROUNDS = 3
WARM = 5
FLOOR_SO = os.path.join(ROOT, 'tools', 'floor', 'libfloor.so')

# case key -> (algorithmic bytes per element, floor it is priced against)
PRODUCT = collections.OrderedDict([
    ('fq_linear_c', (8, 'floor_copy')), ('fq_linear_t', (8, 'floor_copy')),
    ('hist_sym_t_rows', (4, 'floor_read')), ('hist_sym_t_oneshot', (4, 'floor_read')), ('hist_asym_t_oneshot', (4, 'floor_read')),
    ('hist_sym_c', (4, 'floor_read')), ('minmax_t', (4, 'floor_read')), ('minmax_c', (4, 'floor_read')),
    ('quantile_t_hinted', (4, 'floor_read')),
])
FLOORS = collections.OrderedDict([('floor_empty', 0), ('floor_read', 4), ('floor_copy', 8), ('floor_read_atomic_shared', 4),
                                  ('floor_read_atomic_own_rows', 4)])


def ensure_floor_lib():
    """tools/floor/libfloor.so: built by __graft_entry__.build(); compiled here when a box has hipcc but not the file."""
    if os.path.exists(FLOOR_SO): return None
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc): return 'tools/floor/libfloor.so is missing and there is no hipcc to build it'
    try:
        subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-o', FLOOR_SO,
                        os.path.join(ROOT, 'tools', 'floor', 'floor_kernels.hip')], check=True, capture_output=True, timeout=300)
    except Exception as e:
        return f'building libfloor.so failed: {type(e).__name__}: {str(getattr(e, "stderr", b"") or e)[-200:]}'
    return None


def load_floor():
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    f = ctypes.CDLL(FLOOR_SO)
    f.floor_marker.argtypes = [ci, vp]
    f.floor_empty.argtypes = [ci, ci, vp]
    f.floor_read.argtypes = [vp, i64, vp, ci, ci, ci, ci, vp]
    f.floor_copy.argtypes = [vp, vp, i64, ci, ci, ci, vp]
    f.floor_read_atomic.argtypes = [vp, i64, vp, ci, ci, ci, ci, vp]
    return f


class Cases:
    """The case list for one size multiplier: `cases` = [(key, fn(stream))]; every fn makes exactly one library / floor call."""

    def __init__(self, mult, bins, rotate=6):
        import torch
        from ppq_amd import CUDA
        from ppq_amd._lib import lib
        dev = 'cuda'
        self.torch, self.mult = torch, mult
        shp = (B_SHAPE[0] * mult,) + B_SHAPE[1:]
        g = torch.Generator(device=dev).manual_seed(7)
        xs = [torch.randn(*shp, device=dev, generator=g) for _ in range(rotate)]
        outs = [torch.empty_like(xs[0]) for _ in range(rotate)]
        self.n = n = xs[0].numel()
        C, epc = shp[1], shp[2] * shp[3]
        k = [0]

        def nxt():
            k[0] += 1
            return k[0] % rotate
        P = lambda t: t.data_ptr()      # noqa: E731
        sc = torch.rand(C, device=dev, generator=g) * 0.05 + 0.01
        oc = torch.randint(0, 255, [C], device=dev, generator=g).float()
        s1 = torch.tensor([0.03], device=dev); o1 = torch.zeros(1, device=dev)
        hist = torch.zeros(bins, dtype=torch.int32, device=dev)
        hist_c = torch.zeros(C, bins, dtype=torch.int32, device=dev)
        rows = torch.zeros(CUDA.hist_rows(), bins, dtype=torch.int32, device=dev)
        slots = torch.tensor([float('inf'), float('-inf')], device=dev).repeat(CUDA.minmax_slots(), 1).contiguous()
        mins = torch.full([C], float('inf'), device=dev); maxs = torch.full([C], float('-inf'), device=dev)
        amax = float(xs[0].abs().max()); lo, hi = float(xs[0].min()), float(xs[0].max())
        hs = amax / bins
        ws = torch.empty(int(lib.ppqhip_hist_workspace_bytes(n, bins)) + 64, dtype=torch.uint8, device=dev)
        qws = torch.empty(int(lib.ppqhip_quantile_workspace_bytes(n)) + 64, dtype=torch.uint8, device=dev)
        qdest = torch.empty(2, device=dev); qhint = torch.zeros(8, dtype=torch.int32, device=dev)
        sink = torch.zeros(1 << 20, device=dev)
        arows = torch.zeros(256 * bins, dtype=torch.int32, device=dev)
        self.keep = (xs, outs, sc, oc, s1, o1, hist, hist_c, rows, slots, mins, maxs, ws, qws, qdest, qhint, sink, arows)
        fl = self.fl = load_floor()

        def cp(st):
            i = nxt(); return fl.floor_copy(P(xs[i]), P(outs[i]), n, 256, 2, 0, st)
        self.warm_copy = cp
        self.cases = [('floor_empty', lambda st: fl.floor_empty(256, 256, st)),
                       ('floor_read', lambda st: fl.floor_read(P(xs[nxt()]), n, P(sink), 0, 256, 2, 0, st)),
                       ('floor_copy', cp)]
        if mult == 1:       # the histogram's cross-workgroup combine, alone behind a read (grid = the one-shot / rows launch's)
            self.cases += [('floor_read_atomic_shared', lambda st: fl.floor_read_atomic(P(xs[nxt()]), n, P(arows), bins, 98, 0, 9, st)),
                           ('floor_read_atomic_own_rows', lambda st: fl.floor_read_atomic(P(xs[nxt()]), n, P(arows), bins, 256, 2, 9, st))]
        def fq_c(st):
            i = nxt(); return lib.ppqhip_fq_linear_c(P(xs[i]), P(sc), P(oc), P(outs[i]), n, C, epc, 0, 255, 0, st)

        def fq_t(st):
            i = nxt(); return lib.ppqhip_fq_linear_t(P(xs[i]), P(s1), P(o1), P(outs[i]), n, -128, 127, 0, st)
        self.cases += [
            ('fq_linear_c', fq_c), ('fq_linear_t', fq_t),
            ('hist_sym_t_rows', lambda st: lib.ppqhip_hist_sym_t_rows(P(xs[nxt()]), n, hs, 1, P(rows), bins, st)),
            ('hist_sym_t_oneshot', lambda st: lib.ppqhip_hist_sym_t(P(xs[nxt()]), n, hs, 1, P(hist), bins, P(ws), st)),
            ('hist_asym_t_oneshot', lambda st: lib.ppqhip_hist_asym_t(P(xs[nxt()]), n, lo, hi, 1, P(hist), bins, P(ws), st)),
            ('hist_sym_c', lambda st: lib.ppqhip_hist_sym_c(P(xs[nxt()]), n, C, epc, hs, 1, P(hist_c), bins, st)),
            ('minmax_t', lambda st: lib.ppqhip_minmax_t_slots(P(xs[nxt()]), n, P(slots), st)),
            ('minmax_c', lambda st: lib.ppqhip_minmax_c(P(xs[nxt()]), n, C, epc, P(mins), P(maxs), st)),
            ('quantile_t_hinted', lambda st: lib.ppqhip_quantile_t(P(xs[nxt()]), n, 0.9999, P(qdest), P(qhint), P(qws), st)),
        ]


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def check(rc, key):
    if rc not in (0, None):
        from ppq_amd._lib import last_error
        raise RuntimeError(f'{key}: launch failed rc={rc} {last_error()}')


# ---- trace child -------------------------------------------------------------------------------------------------------------------
def trace_child(sizes, bins, manifest_path):
    import torch
    manifest = {'cases': {}, 'iters': {}, 'warm': WARM, 'rounds': ROUNDS}
    cid = 1000
    for mult in sizes:
        cs = Cases(mult, bins)
        it = ITERS.get(mult, 30)
        if mult == sizes[0]:
            for _ in range(3000): cs.warm_copy(stream())          # clocks / caches in a steady state before the first case
        torch.cuda.synchronize()
        ids = {}
        for key, _ in cs.cases:
            ids[key] = cid; manifest['cases'][str(cid)] = [mult, key]; manifest['iters'][str(cid)] = it
            cid += 1
        for _ in range(ROUNDS):           # interleaved rounds: a slow stretch does not own one case
            for key, fn in cs.cases:
                torch.cuda.synchronize()
                cs.fl.floor_marker(ids[key], stream())
                for _ in range(WARM + it): check(fn(stream()), key)
                torch.cuda.synchronize()
        cs.fl.floor_marker(999, stream())
        torch.cuda.synchronize()
        del cs
        torch.cuda.empty_cache()
    json.dump(manifest, open(manifest_path, 'w'))


def parse_trace(trace_csv, manifest):
    """{mult: {key: {'us', 'p10_us', 'kernels'}}}: per call, the sum over its kernels of the median (p10) device duration."""
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    cases = {int(k): v for k, v in manifest['cases'].items()}
    warm = manifest['warm']
    pooled, cur, cur_id = {}, None, None
    for r in rows:
        nm = r['Kernel_Name']
        if 'floor_marker_kernel' in nm:
            gs, wg = int(r['Grid_Size_X']), max(1, int(r['Workgroup_Size_X']))
            cid = gs // wg
            cur, cur_id = (collections.OrderedDict(), cid) if cid in cases else (None, None)
            if cur is not None: pooled.setdefault(cid, []).append(cur)
            continue
        if cur is None: continue
        name = nm.replace('void ', '').replace('ppqhip::', '').split('(')[0]
        cur.setdefault(name, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    out = {}
    for cid, segs in pooled.items():
        mult, key = cases[cid]
        it = manifest['iters'][str(cid)]
        per_kernel = collections.OrderedDict()
        for seg in segs:
            for name, v in seg.items():
                calls = len(v) / float(it + warm)              # launches of this kernel per call of the case
                per_kernel.setdefault(name, [calls, []])[1].extend(v[int(round(warm * calls)):])
        med = p10 = 0.0
        kernels = []
        for name, (calls, v) in per_kernel.items():
            if calls < 0.5 or not v: continue
            v.sort()
            c = max(1, int(round(calls)))
            med += v[len(v) // 2] * c / 1e3; p10 += v[len(v) // 10] * c / 1e3
            kernels.append(f'{name[:48]} x{c}')
        if kernels: out.setdefault(mult, {})[key] = {'us': round(med, 2), 'p10_us': round(p10, 2), 'kernels': kernels}
    return out


def run_trace(sizes, bins, timeout_s=300.0):
    """(results | None, status string).  Never raises; the status says what went wrong (stderr tail of the child included)."""
    rocprof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if rocprof is None: return None, 'rocprofv3 not found'
    work = tempfile.mkdtemp(prefix='ppq_b_', dir='/tmp')
    man = os.path.join(work, 'manifest.json')
    try:
        cmd = [rocprof, '--output-format', 'csv', '--kernel-trace', '-d', work, '-o', 'b', '--', sys.executable, os.path.abspath(__file__),
               '--mode', 'trace-child', '--sizes', ','.join(str(s) for s in sizes), '--bins', str(bins), '--manifest', man]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), capture_output=True, text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired as e:
            tail = (e.stderr.decode('utf-8', 'replace') if isinstance(e.stderr, bytes) else (e.stderr or ''))[-300:]
            return None, f'timeout after {timeout_s:.0f} s; stderr tail: {tail!r}'
        wall = time.perf_counter() - t0
        if not os.path.exists(man):
            return None, f'child rc={r.returncode} wrote no manifest in {wall:.0f} s; stderr tail: {(r.stderr or r.stdout)[-400:]!r}'
        files = glob.glob(os.path.join(work, '**', '*kernel_trace.csv'), recursive=True)
        if not files:
            return None, f'child rc={r.returncode} ok but rocprofv3 wrote no kernel_trace.csv; stderr tail: {(r.stderr or "")[-300:]!r}'
        res = parse_trace(files[0], json.load(open(man)))
        if not res: return None, 'kernel trace parsed to nothing (no marker kernels found)'
        return res, f'ok ({wall:.0f} s)'
    except Exception as e:      # parse errors included: the status carries them
        return None, f'{type(e).__name__}: {e}'
    finally:
        shutil.rmtree(work, ignore_errors=True)


# ---- graph mode (no profiler) --------------------------------------------------------------------------------------------------------
def run_graph(sizes, bins):
    """{mult: {key: {'us'}}}: per-launch time of `iters` launches replayed from one HIP graph (best of 3 replays)."""
    import torch
    out, errors = {}, {}
    for mult in sizes:
        cs = Cases(mult, bins)
        it = ITERS.get(mult, 30)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            if mult == sizes[0]:
                for _ in range(2000): cs.warm_copy(stream())
            for key, fn in cs.cases:                      # eager warm-up on the capture stream (scratch, hints, code objects)
                for _ in range(3): check(fn(stream()), key)
        torch.cuda.synchronize()
        for key, fn in cs.cases:
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(it): check(fn(stream()), key)
                g.replay(); torch.cuda.synchronize()
                best = None
                for _ in range(3):
                    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
                    us = a.elapsed_time(b) / it * 1e3
                    best = us if best is None else min(best, us)
                out.setdefault(mult, {})[key] = {'us': round(best, 2)}
                del g
            except Exception as e:
                errors[f'{key}@x{mult}'] = f'{type(e).__name__}: {str(e)[:120]}'
                torch.cuda.synchronize()
        del cs
        torch.cuda.empty_cache()
    return out, errors


# ---- post-processing -----------------------------------------------------------------------------------------------------------------
def annotate(res):
    """adds GBps / frac (algorithmic bytes over the time) and over_floor to every product case."""
    n0 = B_SHAPE[0] * B_SHAPE[1] * B_SHAPE[2] * B_SHAPE[3]
    for mult, cases in res.items():
        n = n0 * int(mult)
        for key, r in cases.items():
            bpe = PRODUCT[key][0] if key in PRODUCT else FLOORS.get(key, 0)
            if bpe and r['us'] > 0:
                r['GBps'] = round(bpe * n / r['us'] / 1e3, 1)
                r['frac'] = round(bpe * n / r['us'] / 1e3 / HBM_PEAK_GBPS, 3)
            if key in PRODUCT and PRODUCT[key][1] in cases and cases[PRODUCT[key][1]]['us'] > 0:
                r['over_floor'] = round(r['us'] / cases[PRODUCT[key][1]]['us'], 2)
    return res


def measure(bins=2048, sizes=SIZES, trace=True, graph=True, timeout_s=300.0):
    """What bench.py calls.  {'status', 'method', 'trace': {...} | None, 'graph': {...} | None, 'graph_errors'}."""
    err = ensure_floor_lib()
    if err is not None: return {'status': err, 'method': None, 'trace': None, 'graph': None}
    out = {'trace': None, 'graph': None, 'graph_errors': None}
    status = 'trace not requested'
    if trace:
        res, status = run_trace(sizes, bins, timeout_s)
        if res is None:                # one retry on the smallest job: B alone
            res, status2 = run_trace(sizes[:1], bins, timeout_s / 2)
            status = f'{status} | retry with B only: {status2}'
        out['trace'] = annotate(res) if res else None
    if graph:
        try:
            res, errors = run_graph(sizes, bins)
            out['graph'], out['graph_errors'] = annotate(res), (errors or None)
        except Exception as e:
            out['graph_errors'] = {'all': f'{type(e).__name__}: {e}'}
    out['status'] = status
    out['method'] = 'rocprofv3 --kernel-trace medians' if out['trace'] else ('HIP-graph replay / iters (upper bound)' if out['graph'] else None)
    return out


def flat_scalars(m):
    """The flat `config` scalars of the driver's line.  B_* = the tensor the north star names; <kernel>_frac_x<k> = the same
    launch on B x k; <kernel>_crosses_0p70_at_x = the smallest measured multiple at which it reaches 0.70 of 8 TB/s."""
    s = {'B_status': m.get('status'), 'B_method': m.get('method')}
    if m.get('graph_errors'): s['B_graph_errors'] = json.dumps(m['graph_errors'])[:300]
    tr, gr = m.get('trace') or {}, m.get('graph') or {}
    best = tr or gr
    get = lambda d, mult: d.get(mult) or d.get(str(mult)) or {}     # noqa: E731
    b1, g1 = get(tr, 1), get(gr, 1)
    for key in FLOORS:
        short = key.replace('floor_', '')
        if key in b1: s[f'B_floor_{short}_us'] = b1[key]['us']
        if key in g1: s[f'B_floor_{short}_graph_us'] = g1[key]['us']
    for key in PRODUCT:
        if key in b1:
            r = b1[key]
            s[f'B_{key}_rocprof_median_us'] = r['us']; s[f'B_{key}_rocprof_p10_us'] = r['p10_us']
            s[f'B_{key}_frac_of_8TBps'] = r.get('frac'); s[f'B_{key}_over_floor'] = r.get('over_floor')
        if key in g1:
            s[f'B_{key}_graph_us'] = g1[key]['us']; s[f'B_{key}_graph_over_floor'] = g1[key].get('over_floor')
            if key not in b1: s[f'B_{key}_frac_of_8TBps_graph_bound'] = g1[key].get('frac')
        crossed = None
        for mult in sorted(int(k) for k in best):
            r = get(best, mult).get(key)
            if r is None or r.get('frac') is None: continue
            if mult > 1: s[f'{key}_frac_x{mult}'] = r['frac']
            if crossed is None and r['frac'] >= 0.70: crossed = mult
        if any(key in get(best, mu) for mu in best): s[f'{key}_crosses_0p70_at_x'] = crossed
    return s


def report(m, path):
    lines = ['# tools/north_star.py: single-tensor launches on B = [1,512,56,56] fp32 x {multiples}, and their floors; us per call',
             f'# status: {m.get("status")}; graph errors: {m.get("graph_errors")}',
             '# trace = rocprofv3 --kernel-trace median (p10) device duration per call; graph = HIP-graph replay / iters (upper bound: includes the inter-node gap)',
             '# frac = algorithmic bytes (SURVEY 8d: fq 8 B/elem, hist / minmax / quantile 4 B/elem) / time / 8 TB/s; over_floor = time / floor time (copy floor for fq, read floor otherwise)',
             f'# {"size":>5s} {"case":28s} {"trace us":>9s} {"p10":>8s} {"frac":>6s} {"/floor":>6s} {"graph us":>9s} {"g.frac":>6s} {"g./floor":>8s}  kernels']
    tr, gr = m.get('trace') or {}, m.get('graph') or {}
    for mult in sorted({int(k) for k in tr} | {int(k) for k in gr}):
        a, b = tr.get(mult) or tr.get(str(mult)) or {}, gr.get(mult) or gr.get(str(mult)) or {}
        for key in list(FLOORS) + list(PRODUCT):
            ra, rb = a.get(key), b.get(key)
            if ra is None and rb is None: continue
            f = lambda r, k, w, p: (f'{r[k]:{w}.{p}f}' if r and r.get(k) is not None else ' ' * w)      # noqa: E731
            lines.append(f'  x{mult:<4d} {key:28s} {f(ra, "us", 9, 2)} {f(ra, "p10_us", 8, 2)} {f(ra, "frac", 6, 3)} {f(ra, "over_floor", 6, 2)} '
                         f'{f(rb, "us", 9, 2)} {f(rb, "frac", 6, 3)} {f(rb, "over_floor", 8, 2)}  {", ".join((ra or {}).get("kernels", []))}')
    cross = {k: v for k, v in flat_scalars(m).items() if k.endswith('_crosses_0p70_at_x')}
    lines.append('# smallest measured multiple of B at which the launch reaches 0.70 of 8 TB/s (None = not by x32): ' + json.dumps(cross))
    open(path, 'w').write('\n'.join(lines) + '\n')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='both', choices=['trace-child', 'trace', 'graph', 'both'])
    ap.add_argument('--sizes', default=','.join(str(s) for s in SIZES))
    ap.add_argument('--bins', type=int, default=2048)
    ap.add_argument('--manifest', default='')
    ap.add_argument('--report', default='')
    ap.add_argument('--json', default='')
    args = ap.parse_args()
    sizes = tuple(int(v) for v in args.sizes.split(',') if v)
    if args.mode == 'trace-child':
        return trace_child(sizes, args.bins, args.manifest)
    m = measure(args.bins, sizes, trace=args.mode in ('trace', 'both'), graph=args.mode in ('graph', 'both'))
    if args.report: report(m, args.report)
    if args.json: json.dump(m, open(args.json, 'w'), indent=1)
    print(json.dumps(flat_scalars(m), indent=1))


if __name__ == '__main__':
    main()
