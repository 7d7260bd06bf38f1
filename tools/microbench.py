"""Kernel micro-benchmark: GB/s of every hot kernel on A=[1,3,224,224], B=[1,512,56,56], Bx32, through the C ABI.

Honest-bandwidth rules (the round-1 tool rotated inputs only, and `torch.empty_like` handed
every launch the SAME 205 MB output block, which the 256 MiB Infinity Cache can absorb):
  * inputs AND outputs rotate over `--rotate` distinct preallocated buffers per role; at Bx32 the default
    (6) touches 1.2 GB of inputs and 1.2 GB of outputs between two uses of the same buffer;
  * outputs are preallocated and passed to the C entry points (no allocator in the timed loop);
  * every row is priced against BOTH the 8 TB/s HBM3E spec and the 6.29 TB/s float4-copy ceiling that
    MI355X_MICROARCH.md measures, next to a plain `out.copy_(x)` over the same rotating buffers.
Timing: torch.cuda.Event pairs on the current stream (the stream the library launches on), `iters`
back-to-back launches per measurement -> average launch-to-launch time (includes the inter-kernel gap)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd import CUDA  # noqa: E402
from ppq_amd._lib import lib  # noqa: E402

SPEC, COPY = 8000.0, 6290.0


def stream():
    return torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=50, warmup=5):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--relu', action='store_true')
    ap.add_argument('--relu6', action='store_true', help='clamp(4 x, 0, 6): ties on both ends')
    ap.add_argument('--bins', type=int, default=2048)
    ap.add_argument('--only', type=str, default='')
    ap.add_argument('--rotate', type=int, default=6, help='distinct buffers per role (inputs, outputs)')
    ap.add_argument('--tensors', type=str, default='A,B,Bx32')
    args = ap.parse_args()
    dev = 'cuda'
    torch.manual_seed(0)
    shapes = {'A': (1, 3, 224, 224), 'B': (1, 512, 56, 56), 'Bx8': (8, 512, 56, 56), 'Bx32': (32, 512, 56, 56)}
    rows = []
    for name, shp in shapes.items():
        if name not in args.tensors.split(','): continue
        R = args.rotate
        xs = [torch.randn(*shp, device=dev) for _ in range(R)]
        if args.relu: xs = [torch.relu(t) for t in xs]
        if args.relu6: xs = [torch.clamp(4 * t, 0, 6) for t in xs]
        dys = [torch.rand_like(xs[0]) for _ in range(R)]
        outs = [torch.empty_like(xs[0]) for _ in range(R)]
        n, C = xs[0].numel(), shp[1]
        epc = n // (shp[0] * C)
        k = [0]

        def nxt():
            k[0] += 1
            return k[0] % R
        s1 = torch.tensor([0.03], device=dev); o1 = torch.zeros(1, device=dev)
        sc = torch.rand(C, device=dev) * 0.05 + 0.01; oc = torch.randint(0, 255, [C], device=dev).float()
        gs1 = torch.zeros(1, device=dev); gsc = torch.zeros(C, device=dev)
        hist = torch.zeros(args.bins, dtype=torch.int32, device=dev)
        rowsbuf = torch.zeros(CUDA.hist_rows(), args.bins, dtype=torch.int32, device=dev)
        hist_c = torch.zeros(C, args.bins, dtype=torch.int32, device=dev)
        slots = torch.tensor([float('inf'), float('-inf')], device=dev).repeat(CUDA.minmax_slots(), 1).contiguous()
        mins = torch.full([C], float('inf'), device=dev); maxs = torch.full([C], float('-inf'), device=dev)
        hs = float(xs[0].abs().max()) / args.bins
        ws = torch.empty(int(lib.ppqhip_hist_workspace_bytes(n, args.bins)) + 64, dtype=torch.uint8, device=dev)

        def P(t): return t.data_ptr()
        scp2 = torch.full([C], 2.0 ** -5, device=dev); ocz = torch.zeros(C, device=dev)
        iso4 = torch.zeros(4, device=dev); csum = torch.zeros(C, dtype=torch.float64, device=dev)
        qws = torch.empty(int(lib.ppqhip_quantile_workspace_bytes(n)) + 64, dtype=torch.uint8, device=dev)
        from ppq_amd.ffi import quantile_hint
        qhint = quantile_hint(torch.device(dev))

        def fq_t():
            i = nxt(); lib.ppqhip_fq_linear_t(P(xs[i]), P(s1), P(o1), P(outs[i]), n, -128, 127, 0, stream())

        def fq_c():
            i = nxt(); lib.ppqhip_fq_linear_c(P(xs[i]), P(sc), P(oc), P(outs[i]), n, C, epc, 0, 255, 0, stream())

        sp2 = torch.tensor([2.0 ** -5], device=dev)         # FP8 scales are powers of two in PPQ (observer/floating.py:97)

        def fq_f():
            i = nxt(); lib.ppqhip_fq_float_t(P(xs[i]), P(sp2), P(o1), P(outs[i]), n, 4, 3, -448.0, 448.0, 0, stream())

        def fq_f_generic():
            i = nxt(); lib.ppqhip_fq_float_t(P(xs[i]), P(s1), P(o1), P(outs[i]), n, 4, 3, -448.0, 448.0, 0, stream())

        def bwd_t():
            i = nxt(); lib.ppqhip_fq_linear_t_bwd(P(xs[i]), P(s1), P(o1), P(dys[i]), P(outs[i]), P(gs1), n, -128, 127, 0, stream())

        def bwd_c():
            i = nxt(); lib.ppqhip_fq_linear_c_bwd(P(xs[i]), P(sc), P(oc), P(dys[i]), P(outs[i]), P(gsc), n, C, epc, 0, 255, 0, stream())

        def copy():
            i = nxt(); outs[i].copy_(xs[i])
        cases = {
            'fq_linear_t': (8, fq_t), 'fq_linear_c': (8, fq_c), 'fq_float_t (E4M3, s=2^-5)': (8, fq_f), 'fq_float_t (generic s)': (8, fq_f_generic),
            'hist_sym_t (one-shot)': (4, lambda: lib.ppqhip_hist_sym_t(P(xs[nxt()]), n, hs, 1, P(hist), args.bins, P(ws), stream())),
            'hist_sym_t (rows)': (4, lambda: lib.ppqhip_hist_sym_t_rows(P(xs[nxt()]), n, hs, 1, P(rowsbuf), args.bins, stream())),
            'hist_sym_c (per channel)': (4, lambda: lib.ppqhip_hist_sym_c(P(xs[nxt()]), n, C, epc, hs, 1, P(hist_c), args.bins, stream())),
            'minmax_t (slots)': (4, lambda: lib.ppqhip_minmax_t_slots(P(xs[nxt()]), n, P(slots), stream())),
            'minmax_c': (4, lambda: lib.ppqhip_minmax_c(P(xs[nxt()]), n, C, epc, P(mins), P(maxs), stream())),
            'quantile_t (hinted)': (4, lambda: CUDA.Quantile_Hinted(xs[nxt()], 0.9999, qhint)),   # an observer's call: thresholds of the previous batch
            'quantile_t (cold)': (4, lambda: CUDA.Quantile_Hinted(xs[nxt()], 0.9999, None)),      # every call samples its thresholds
            'lsq_bwd_t': (12, bwd_t), 'lsq_bwd_c': (12, bwd_c),
            'fq_float_c (E4M3, per channel)': (8, lambda: lib.ppqhip_fq_float_c(P(xs[nxt()]), P(scp2), P(ocz), P(outs[nxt()]), n, C, epc, 4, 3, -448.0, 448.0, 0, stream())),
            'fq_float_c_bwd': (12, lambda: lib.ppqhip_fq_float_c_bwd(P(xs[nxt()]), P(scp2), P(ocz), P(dys[nxt()]), P(outs[nxt()]), P(gsc), n, C, epc, 4, 3, -448.0, 448.0, 0, stream())),
            'to_int_t (int8 out)': (5, lambda: lib.ppqhip_to_int_t(P(xs[nxt()]), P(s1), P(o1), P(outs[nxt()]), n, -128, 127, 0, 0, stream())),
            'to_int_c (int32 out)': (8, lambda: lib.ppqhip_to_int_c(P(xs[nxt()]), P(sc), P(oc), P(outs[nxt()]), n, C, epc, -32768, 32767, 0, 2, stream())),
            'isotone_t': (4, lambda: lib.ppqhip_isotone_t(P(xs[nxt()]), n, P(iso4), P(qws), stream())),
            'channel_sum': (4, lambda: lib.ppqhip_channel_sum(P(xs[nxt()]), n, C, epc, P(csum), stream())),
            'torch out.copy_(x) (ref)': (8, copy),
            'torch abs().max() (ref)': (4, lambda: xs[nxt()].abs().max()),
        }
        for kname, (bpe, fn) in cases.items():
            if args.only and not any(o in kname for o in args.only.split(',')): continue
            t = timeit(fn, iters=200 if n < 10_000_000 else 30)
            gbps = bpe * n / t / 1e9
            rows.append({'kernel': kname, 'tensor': name, 'us': round(t * 1e6, 2), 'GBps': round(gbps, 1),
                         'frac_of_8TBps': round(gbps / SPEC, 3), 'frac_of_copy_ceiling': round(gbps / COPY, 3)})
            print(f'{kname:26s} {name:5s} {t*1e6:10.2f} us  {gbps:9.1f} GB/s  {100*gbps/SPEC:5.1f} % of 8 TB/s  '
                  f'{100*gbps/COPY:6.1f} % of 6.29 TB/s', flush=True)
        del xs, dys, outs
        torch.cuda.empty_cache()
    os.makedirs('gpurun_out', exist_ok=True)
    tag = ('relu6' if args.relu6 else 'relu' if args.relu else 'randn') + f'_{args.bins}'
    json.dump(rows, open(f'gpurun_out/microbench_{tag}.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
