"""A/B measurement of hist.hip builds (tools/build_hist_variants.sh -> variants/lib_*.so) in ONE process.

Workloads: (1) the in-situ launch of bench.py -- the 72 activation tensors one ResNet-50 calibration
forward at batch 32 observes (2.149 GB, real ReLU / residual value distributions, real hist scales),
binned by ONE ppqhip_hist_t_rows_multi launch; (2) single tensors A, B, Bx32 (randn and relu, inputs
rotated so the 256 MiB Infinity Cache cannot serve them).  Every variant's counts are compared with the
first library's (round 1's kernel, itself exact against the oracle): identical or the row says MISMATCH.
Timing: hipEvent pairs on torch's current stream around `iters` back-to-back launches."""
import ctypes
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppq_amd  # noqa: E402  (loads torch's HIP runtime first)
from ppq_amd import ffi  # noqa: E402

HIST_JOB = np.dtype([('x', '<u8'), ('rows', '<u8'), ('n', '<i8'), ('p0', '<f4'), ('p1', '<f4')])
c_vp, c_i64, c_int, c_flt = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float


class Lib:
    def __init__(self, path):
        self.name = os.path.basename(path)[4:-3]
        self.l = ctypes.CDLL(path)
        self.l.ppqhip_hist_rows.restype = c_i64
        self.l.ppqhip_hist_t_rows_multi.argtypes = [c_vp, c_int, c_int, c_int, c_i64, c_vp]
        self.l.ppqhip_hist_rows_finish.argtypes = [c_vp, c_i64, c_vp, c_vp]
        self.l.ppqhip_hist_sym_t.argtypes = [c_vp, c_i64, c_flt, c_int, c_vp, c_i64, c_vp, c_vp]
        self.l.ppqhip_hist_asym_t.argtypes = [c_vp, c_i64, c_flt, c_flt, c_int, c_vp, c_i64, c_vp, c_vp]
        self.l.ppqhip_hist_workspace_bytes.restype = c_i64
        self.l.ppqhip_hist_workspace_bytes.argtypes = [c_i64, c_i64]
        self.l.ppqhip_last_error.restype = ctypes.c_char_p
        self.R = int(self.l.ppqhip_hist_rows())

    def check(self, st):
        if st != 0: raise RuntimeError(f'{self.name}: {self.l.ppqhip_last_error()}')


def stream():
    return torch.cuda.current_stream().cuda_stream


def timeit(fn, iters, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def capture_insitu(batch=32):
    """The (tensors, hist scales) of the phase-2 statistics launch of one calibration forward."""
    sys.argv = ['bench.py']
    import bench
    from ppq_amd.calibration import RuntimeCalibrationPass
    graph, ex = bench.build_workload('cuda:0', 2048, 'kl')
    g = torch.Generator(device='cuda').manual_seed(1234)
    batches = [torch.rand(batch, 3, 224, 224, device='cuda', generator=g)]
    got = {}
    orig = ffi.CUDA.Histogram_T_Rows_Multi

    def spy(tensors, rows, scales, clip_outliers=True):
        got['x'] = [t.detach().clone() for t in tensors]; got['hs'] = [float(s) for s in scales]
        return orig(tensors, rows, scales, clip_outliers)
    ffi.CUDA.Histogram_T_Rows_Multi = spy          # class attribute: observer.py calls it through the class
    p = RuntimeCalibrationPass(method='kl', check_steps=False, use_hip_graph=False)
    p.optimize(graph, dataloader=batches, executor=ex, calib_steps=1)
    ffi.CUDA.Histogram_T_Rows_Multi = orig
    torch.cuda.synchronize()
    del graph, ex, p
    torch.cuda.empty_cache()
    return got['x'], got['hs']


def main():
    only = sys.argv[1].split(',') if len(sys.argv) > 1 else None
    paths = sorted(glob.glob(os.path.join(ROOT, 'variants', 'lib_*.so')))
    paths.sort(key=lambda p: (not p.endswith('lib_r01.so'), p))
    libs = [Lib(p) for p in paths if only is None or any(o in p for o in only) or p.endswith('lib_r01.so')]
    print('variants:', [l.name for l in libs], flush=True)
    bins = 2048
    out = {}
    # ---- (1) in situ
    if os.environ.get('HV_SKIP_INSITU'): xs, hs = [], []
    else: xs, hs = capture_insitu()
    total = sum(x.numel() for x in xs) * 4
    print(f'in-situ set: {len(xs)} tensors, {total/1e9:.3f} GB', flush=True)
    ref_hists = None
    for lib in (libs if xs else []):
        rows = [torch.zeros(lib.R, bins, dtype=torch.int32, device='cuda') for _ in xs]
        jobs = np.empty(len(xs), dtype=HIST_JOB)
        jobs['x'] = [x.data_ptr() for x in xs]; jobs['rows'] = [r.data_ptr() for r in rows]
        jobs['n'] = [x.numel() for x in xs]; jobs['p0'] = np.asarray(hs, dtype=np.float32); jobs['p1'] = 0

        def launch():
            lib.check(lib.l.ppqhip_hist_t_rows_multi(jobs.ctypes.data, len(xs), 0, 1, bins, stream()))
        launch(); torch.cuda.synchronize()
        hists = torch.zeros(len(xs), bins, dtype=torch.int32, device='cuda')
        for k, r in enumerate(rows): lib.check(lib.l.ppqhip_hist_rows_finish(r.data_ptr(), bins, hists[k].data_ptr(), stream()))
        torch.cuda.synchronize()
        if ref_hists is None: ref_hists = hists.clone(); ok = 'ref'
        else: ok = 'same' if torch.equal(hists, ref_hists) else f'MISMATCH({int((hists != ref_hists).sum())} bins)'
        us = min(timeit(launch, 10) for _ in range(3))
        out[f'{lib.name}/insitu'] = us
        print(f'{lib.name:22s} insitu 72x  {us:9.1f} us  {total/us/1e6:6.3f} TB/s  frac {total/us/1e6/8:.3f}  [{ok}]', flush=True)
        del rows
    del xs
    torch.cuda.empty_cache()
    # ---- (2) single tensors
    torch.manual_seed(0)
    shapes = {'A': (1, 3, 224, 224), 'B': (1, 512, 56, 56), 'Bx4': (4, 512, 56, 56), 'Bx32': (32, 512, 56, 56)}
    if os.environ.get('HV_TENSORS'): shapes = {k: v for k, v in shapes.items() if k in os.environ['HV_TENSORS'].split(',')}
    for tname, shp in shapes.items():
        for dist in ('randn', 'relu'):
            rot = 4 if tname == 'Bx32' else 1
            ts = [torch.randn(*shp, device='cuda') for _ in range(rot)]
            if dist == 'relu': ts = [torch.relu(t) for t in ts]
            n = ts[0].numel()
            hscale = float(ts[0].abs().max()) / bins
            lo, hi = float(ts[0].min()), float(ts[0].max())
            ref = {}
            for lib in libs:
                ws = torch.empty(max(int(lib.l.ppqhip_hist_workspace_bytes(n, bins)), 16), dtype=torch.uint8, device='cuda')
                for kind in ('sym', 'asym'):
                    hist = torch.zeros(bins, dtype=torch.int32, device='cuda')
                    cnt = [0]

                    def launch():
                        cnt[0] += 1
                        x = ts[cnt[0] % rot]
                        if kind == 'sym':
                            lib.check(lib.l.ppqhip_hist_sym_t(x.data_ptr(), n, hscale, 1, hist.data_ptr(), bins, ws.data_ptr(), stream()))
                        else:
                            lib.check(lib.l.ppqhip_hist_asym_t(x.data_ptr(), n, lo, hi, 1, hist.data_ptr(), bins, ws.data_ptr(), stream()))
                    cnt[0] = -1; launch(); torch.cuda.synchronize()      # x = ts[0]
                    key = (kind,)
                    if key not in ref: ref[key] = hist.clone(); ok = 'ref'
                    else: ok = 'same' if torch.equal(hist, ref[key]) else f'MISMATCH({int((hist != ref[key]).sum())} bins)'
                    us = min(timeit(launch, 200 if n < 10_000_000 else 20) for _ in range(3))
                    out[f'{lib.name}/{tname}/{dist}/{kind}'] = us
                    print(f'{lib.name:22s} {tname:5s} {dist:5s} {kind:4s} {us:9.2f} us  {4*n/us/1e6:6.3f} TB/s  [{ok}]', flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'hist_variants.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
