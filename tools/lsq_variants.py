"""A/B measurement of linear.hip builds (SRC=linear.hip tools/build_hist_variants.sh lsq_<name>:"-D..." ->
variants/lib_lsq_*.so): LSQ backward per tensor / per channel on Bx32 = [32,512,56,56] (12 B / element: x and dy
read, grad_x written), inputs and outputs rotating over 6 buffers each (3.7 GB between two uses of one buffer)."""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppq_amd  # noqa: E402,F401

c_vp, c_i64, c_int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    paths = sorted(glob.glob(os.path.join(ROOT, 'variants', 'lib_lsq_*.so')))
    shp = (32, 512, 56, 56)
    R = 6
    torch.manual_seed(0)
    xs = [torch.randn(*shp, device='cuda') for _ in range(R)]
    dys = [torch.rand_like(xs[0]) for _ in range(R)]
    outs = [torch.empty_like(xs[0]) for _ in range(R)]
    n, C = xs[0].numel(), shp[1]
    epc = n // (shp[0] * C)
    s1 = torch.tensor([0.03], device='cuda'); o1 = torch.zeros(1, device='cuda'); g1 = torch.zeros(1, device='cuda')
    sc = torch.rand(C, device='cuda') * 0.05 + 0.01; oc = torch.randint(0, 255, [C], device='cuda').float(); gc = torch.zeros(C, device='cuda')
    ref = {}
    for p in paths:
        l = ctypes.CDLL(p)
        l.ppqhip_fq_linear_t_bwd.argtypes = [c_vp] * 6 + [c_i64, c_int, c_int, c_int, c_vp]
        l.ppqhip_fq_linear_c_bwd.argtypes = [c_vp] * 6 + [c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp]
        st = torch.cuda.current_stream().cuda_stream
        k = [0]

        def t():
            k[0] += 1; i = k[0] % R
            l.ppqhip_fq_linear_t_bwd(xs[i].data_ptr(), s1.data_ptr(), o1.data_ptr(), dys[i].data_ptr(), outs[i].data_ptr(), g1.data_ptr(), n, -128, 127, 0, st)

        def c():
            k[0] += 1; i = k[0] % R
            l.ppqhip_fq_linear_c_bwd(xs[i].data_ptr(), sc.data_ptr(), oc.data_ptr(), dys[i].data_ptr(), outs[i].data_ptr(), gc.data_ptr(), n, C, epc, 0, 255, 0, st)
        name = os.path.basename(p)[8:-3]
        for kind, fn, g in (('lsq_bwd_t', t, g1), ('lsq_bwd_c', c, gc)):
            k[0] = -1; fn(); torch.cuda.synchronize()
            sig = (float(outs[0].double().sum()), g.double().sum().item())
            ok = 'ref' if kind not in ref else ('same' if abs(sig[0] - ref[kind][0]) <= 1e-9 * abs(ref[kind][0]) and abs(sig[1] - ref[kind][1]) <= 1e-4 * abs(ref[kind][1]) + 1e-12 else f'DIFF {sig} vs {ref[kind]}')
            ref.setdefault(kind, sig)
            us = min(timeit(fn) for _ in range(3))
            print(f'{name:24s} {kind:10s} {us:8.1f} us  {12 * n / us / 1e6:6.3f} TB/s  frac {12 * n / us / 1e6 / 8:.3f}  [{ok}]', flush=True)


if __name__ == '__main__':
    main()
