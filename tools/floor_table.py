"""Floor table for the single-tensor launches on B = [1,512,56,56] ("Next round" item 1).

Runs, on rotating buffers like tools/microbench.py, (a) the floor kernels of tools/floor/floor_kernels.hip -- an empty kernel, pure
reads, copies, the cross-workgroup atomic combine alone, a returning-atomic ticket -- and (b) the library's own single-tensor
entry points.  Meant to be run UNDER `rocprofv3 --kernel-trace`: every case is preceded by one launch of `floor_marker_kernel`
whose grid size is the case id, and `gpurun_out/floor_manifest_<tag>.json` maps ids to case names; tools/floor_report.py turns
the kernel trace + manifest into per-case device-duration medians.  Without the tracer it still prints launch-to-launch times.

  rocprofv3 --output-format csv --kernel-trace -d gpurun_out/floor_trace -o floor -- python tools/floor_table.py --tag base
  PPQHIP_LIBRARY=variants/lib_x.so rocprofv3 ... -- python tools/floor_table.py --tag x --product-only
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppq_amd import CUDA  # noqa: E402
from ppq_amd._lib import lib  # noqa: E402

vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int


def load_floor():
    path = os.path.join(ROOT, 'tools', 'floor', 'libfloor.so')
    if not os.path.exists(path):
        raise SystemExit(f'{path} is missing: run tools/floor/build.sh')
    f = ctypes.CDLL(path)
    f.floor_marker.argtypes = [ci, vp]
    f.floor_empty.argtypes = [ci, ci, vp]
    f.floor_read.argtypes = [vp, i64, vp, ci, ci, ci, ci, vp]
    f.floor_copy.argtypes = [vp, vp, i64, ci, ci, ci, vp]
    f.floor_atomic.argtypes = [vp, ci, ci, ci, ci, ci, vp]
    f.floor_read_atomic.argtypes = [vp, i64, vp, ci, ci, ci, ci, vp]
    f.floor_ticket.argtypes = [vp, vp, ci, ci, vp]
    return f


def stream():
    return torch.cuda.current_stream().cuda_stream


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', default='base')
    ap.add_argument('--product-only', action='store_true')
    ap.add_argument('--floor-only', action='store_true')
    ap.add_argument('--hist-wg', type=str, default='', help='comma list: sweep PPQHIP_DEV_HIST_WG of the one-shot histogram (needs a -DPPQHIP_DEV_KNOBS build)')
    ap.add_argument('--rows-wg', type=str, default='', help='the same for the persistent-rows entry point')
    ap.add_argument('--libs', type=str, default='', help='tag=path,...: further builds of libppq_hip.so whose product cases are INTERLEAVED with '
                    'the default library in the same process (A/B under identical clocks / memory state)')
    ap.add_argument('--rounds', type=int, default=1, help='repeat the whole case list this many times (the report pools the rounds)')
    ap.add_argument('--cases', type=str, default='', help='comma list of substrings: run only matching product cases')
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--shape', type=str, default='1,512,56,56')
    ap.add_argument('--rotate', type=int, default=6)
    ap.add_argument('--bins', type=int, default=2048)
    args = ap.parse_args()
    fl = load_floor()
    dev = 'cuda'
    torch.manual_seed(0)
    shp = tuple(int(v) for v in args.shape.split(','))
    R = args.rotate
    xs = [torch.randn(*shp, device=dev) for _ in range(R)]
    outs = [torch.empty_like(xs[0]) for _ in range(R)]
    n, C = xs[0].numel(), shp[1]
    epc = n // (shp[0] * C)
    k = [0]

    def nxt():
        k[0] += 1
        return k[0] % R

    def P(t): return t.data_ptr()
    sink = torch.zeros(65536, device=dev)
    bins = args.bins
    arows = torch.zeros(2048 * bins, dtype=torch.int32, device=dev)      # up to 2048 rows for the per-workgroup mode
    cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    s1 = torch.tensor([0.03], device=dev); o1 = torch.zeros(1, device=dev)
    sc = torch.rand(C, device=dev) * 0.05 + 0.01; oc = torch.randint(0, 255, [C], device=dev).float()
    hist = torch.zeros(bins, dtype=torch.int32, device=dev)
    rowsbuf = torch.zeros(CUDA.hist_rows(), bins, dtype=torch.int32, device=dev)
    hist_c = torch.zeros(C, bins, dtype=torch.int32, device=dev)
    slots = torch.tensor([float('inf'), float('-inf')], device=dev).repeat(CUDA.minmax_slots(), 1).contiguous()
    mins = torch.full([C], float('inf'), device=dev); maxs = torch.full([C], float('-inf'), device=dev)
    hs = float(xs[0].abs().max()) / bins
    ws = torch.empty(int(lib.ppqhip_hist_workspace_bytes(n, bins)) + 64, dtype=torch.uint8, device=dev)

    cases = []          # (name, fn)
    if not args.product_only:
        for g, b in ((256, 256), (256, 512), (784, 256), (1568, 256), (98, 512)):
            cases.append((f'floor_empty grid={g} block={b}', lambda g=g, b=b: fl.floor_empty(g, b, stream())))
        for b, u, g in ((512, 2, 98), (512, 2, 196), (512, 2, 256), (512, 2, 0), (512, 1, 0), (256, 2, 0), (256, 1, 0), (256, 2, 256), (256, 4, 0),
                        (1024, 1, 256), (1024, 2, 98), (1024, 1, 0)):
            cases.append((f'floor_read block={b} U={u} grid={g if g else "one tile per WG"}',
                          lambda b=b, u=u, g=g: fl.floor_read(P(xs[nxt()]), n, P(sink), g, b, u, 0, stream())))
        for b, u, st in ((256, 2, 0), (256, 1, 0), (256, 4, 0), (512, 1, 0), (512, 2, 0), (1024, 1, 0), (256, 2, 1), (256, 1, 1)):
            def f(b=b, u=u, st=st):
                i = nxt(); return fl.floor_copy(P(xs[i]), P(outs[i]), n, b, u, st, stream())
            cases.append((f'floor_copy block={b} U={u} store={"nt" if st else "plain"}', f))
        for g, mode, keep in ((98, 0, 9), (98, 1, 9), (98, 2, 9), (256, 0, 9), (256, 1, 9), (256, 2, 9), (8, 0, 16), (32, 0, 9), (98, 0, 16), (98, 0, 4)):
            cases.append((f'floor_atomic grid={g} block=512 rows={("one", "per blockIdx%8", "per WG")[mode]} keep={keep}/16 of {bins}',
                          lambda g=g, mode=mode, keep=keep: fl.floor_atomic(P(arows), bins, g, 512, mode, keep, stream())))
        for g, mode, keep in ((98, 0, 9), (196, 0, 9), (256, 0, 9), (256, 2, 9), (98, 0, 0), (256, 0, 0)):
            cases.append((f'floor_read_atomic grid={g} block=512 U=2 rows={("one", "per blockIdx%8", "per WG")[mode]} keep={keep}/16',
                          lambda g=g, mode=mode, keep=keep: fl.floor_read_atomic(P(xs[nxt()]), n, P(arows), bins, g, mode, keep, stream())))
        for g in (98, 256):
            cases.append((f'floor_ticket grid={g} block=64', lambda g=g: fl.floor_ticket(P(cnt), P(cnt) + 8, g, 64, stream())))

    def product_cases(tag, L, knobs):
        pre = f'lib[{tag}] '

        def fq_t():
            i = nxt(); return L.ppqhip_fq_linear_t(P(xs[i]), P(s1), P(o1), P(outs[i]), n, -128, 127, 0, stream())

        def fq_c():
            i = nxt(); return L.ppqhip_fq_linear_c(P(xs[i]), P(sc), P(oc), P(outs[i]), n, C, epc, 0, 255, 0, stream())
        out = [
            (pre + 'fq_linear_t', fq_t), (pre + 'fq_linear_c', fq_c),
            (pre + 'hist_sym_t (one-shot)', lambda: L.ppqhip_hist_sym_t(P(xs[nxt()]), n, hs, 1, P(hist), bins, P(ws), stream())),
            (pre + 'hist_sym_t (rows)', lambda: L.ppqhip_hist_sym_t_rows(P(xs[nxt()]), n, hs, 1, P(rowsbuf), bins, stream())),
            (pre + 'hist_sym_c', lambda: L.ppqhip_hist_sym_c(P(xs[nxt()]), n, C, epc, hs, 1, P(hist_c), bins, stream())),
            (pre + 'minmax_t (slots)', lambda: L.ppqhip_minmax_t_slots(P(xs[nxt()]), n, P(slots), stream())),
            (pre + 'minmax_c', lambda: L.ppqhip_minmax_c(P(xs[nxt()]), n, C, epc, P(mins), P(maxs), stream())),
        ]
        if knobs:
            def knob_case(env, fn):
                def f():
                    os.environ.update(env)
                    r = fn()
                    for k_ in env: os.environ.pop(k_)
                    return r
                return f
            one = lambda: L.ppqhip_hist_sym_t(P(xs[nxt()]), n, hs, 1, P(hist), bins, P(ws), stream())      # noqa: E731
            rws = lambda: L.ppqhip_hist_sym_t_rows(P(xs[nxt()]), n, hs, 1, P(rowsbuf), bins, stream())      # noqa: E731
            out.append((pre + 'hist_sym_t (one-shot) persistent kernel', knob_case({'PPQHIP_DEV_HIST_SMALL': '0'}, one)))
            out.append((pre + 'hist_sym_t (rows) persistent kernel', knob_case({'PPQHIP_DEV_HIST_SMALL': '0'}, rws)))
            for wg in [w for w in args.hist_wg.split(',') if w]:
                out.append((pre + f'hist_sym_t (one-shot) small kernel grid={wg}', knob_case({'PPQHIP_DEV_HIST_WG': wg}, one)))
            for wg in [w for w in args.rows_wg.split(',') if w]:
                out.append((pre + f'hist_sym_t (rows) small kernel grid={wg}', knob_case({'PPQHIP_DEV_HIST_WG': wg}, rws)))
        return out

    if not args.floor_only:
        from ppq_amd._lib import PROTOTYPES
        libs = [('head', lib, False)]
        for spec in [v for v in args.libs.split(',') if v]:
            tag, path = spec.split('=')
            L = ctypes.CDLL(os.path.join(ROOT, path))
            for nm in ('ppqhip_fq_linear_t', 'ppqhip_fq_linear_c', 'ppqhip_hist_sym_t', 'ppqhip_hist_sym_t_rows', 'ppqhip_hist_sym_c',
                       'ppqhip_minmax_t_slots', 'ppqhip_minmax_c'):
                getattr(L, nm).restype, getattr(L, nm).argtypes = PROTOTYPES[nm]
            libs.append((tag, L, tag.startswith('dev')))
        per_lib = [product_cases(tag, L, knobs) for tag, L, knobs in libs]
        for i in range(max(len(c) for c in per_lib)):                  # interleave: the same case of every library back to back
            for c in per_lib:
                if i < len(c): cases.append(c[i])

        def cp():
            i = nxt(); outs[i].copy_(xs[i]); return 0
        cases.append(('torch out.copy_(x)', cp))
        if args.cases:
            keep = [c for c in args.cases.split(',') if c]
            cases = [c for c in cases if not c[0].startswith('lib[') or any(k_ in c[0] for k_ in keep)]

    for _ in range(3000):                                              # clocks / caches in a steady state before the first case
        i = nxt(); fl.floor_copy(P(xs[i]), P(outs[i]), n, 256, 2, 0, stream())
    torch.cuda.synchronize()

    manifest, rows = {}, []
    for rnd in range(args.rounds):
        for idx, (name, fn) in enumerate(cases):
            cid = 1000 + idx
            manifest[cid] = name
            torch.cuda.synchronize()
            fl.floor_marker(cid, stream())
            for _ in range(5):
                rc = fn()
                if rc not in (0, None):
                    raise SystemExit(f'{name}: launch failed rc={rc} {lib.ppqhip_last_error()}')
            torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters): fn()
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / args.iters * 1e3
            if rnd == 0: rows.append([name, us])
            else: rows[idx][1] = min(rows[idx][1], us)
            print(f'round {rnd} {name:100s} launch-to-launch {us:8.2f} us', flush=True)
    fl.floor_marker(999, stream())
    torch.cuda.synchronize()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump({'shape': shp, 'iters': args.iters, 'warmup': 5, 'rounds': args.rounds, 'cases': manifest, 'launch_to_launch_us': rows,
               'library': os.environ.get('PPQHIP_LIBRARY', 'ppq_amd/libppq_hip.so')},
              open(os.path.join(ROOT, 'gpurun_out', f'floor_manifest_{args.tag}.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
