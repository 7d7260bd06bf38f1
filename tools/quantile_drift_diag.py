import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from ppq_amd import CUDA, ffi, _lib
from ppq_amd.ffi import quantile_hint
import ctypes
dev = torch.device('cuda')
lay = (ctypes.c_int64 * 8)(); _lib.lib.ppqhip_quantile_hot_layout(lay)
off_rec = lay[1]
m = int(sys.argv[1]) if len(sys.argv) > 1 else 8
jit = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
n = m * 512 * 56 * 56
g = torch.Generator(device=dev).manual_seed(0); rng = np.random.default_rng(0)
base = [torch.randn(n, device=dev, generator=g) for _ in range(8)]
scales = np.exp(rng.normal(0, jit, 8))
xs = [b * float(s) for b, s in zip(base, scales)]
hint = quantile_hint(dev)
CUDA.Quantile_Hinted(xs[0], 0.9999, hint); torch.cuda.synchronize()
wanted = n - int(np.rint(np.float32(n) * np.float32(0.9999)))
prev = int(hint.cpu()[7]); last = 0
for c in range(40):
    i = int(rng.integers(0, 8))
    hb = hint.cpu().tolist()
    CUDA.Quantile_Hinted(xs[i], 0.9999, hint); torch.cuda.synchronize()
    ws = list(ffi._workspaces.values())[-1].view(torch.int32)
    rec = ws[off_rec: off_rec + 512 * 16].view(512, 16).cpu().numpy()
    h = hint.cpu().tolist()
    print(f'call {c:2d} scale {scales[i]:.3f} (x{scales[i]/scales[last]:.3f}) level before {(hb[0]>>8)&3}/{(hb[2]>>8)&3} listed hi {rec[:,0].sum():6d} lo {rec[:,8].sum():6d} max/wg {rec[:,0].max():3d} wanted {wanted} settled {h[7]-prev} level after {(h[0]>>8)&3}/{(h[2]>>8)&3} valid {h[0]&255}/{h[2]&255}')
    prev = h[7]; last = i
