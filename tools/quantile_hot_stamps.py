"""Developer aid: phase stamps of the selecting workgroup of quantile_hot_select_kernel (a -DPPQHIP_QH_TIMING build, pointed at with
PPQHIP_LIBRARY): ws[32 + i] = s_memrealtime (100 MHz) at  0 kernel entry, 1 role known (ticket, header, records back),
2 counts scanned, 3 keys gathered in LDS, 4 wave selects done, 6 select returned, 7 results written.
    PPQHIP_LIBRARY=variants/lib_qhtime.so python tools/quantile_hot_stamps.py [sizes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd._lib import lib  # noqa: E402

dev = torch.device('cuda')
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '1,8,32').split(',')]
g = torch.Generator(device=dev).manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for m in sizes:
    n = m * 512 * 56 * 56
    xs = [torch.randn(n, device=dev, generator=g) for _ in range(3)]
    ws = torch.zeros(int(lib.ppqhip_quantile_workspace_bytes(n)) // 4 + 16, dtype=torch.int32, device=dev)
    hint = torch.zeros(8, dtype=torch.int32, device=dev); dest = torch.zeros(2, device=dev)
    rows, mhz, inner = [], [], []
    for i in range(40):
        rc = lib.ppqhip_quantile_t(xs[i % 3].data_ptr(), n, 0.9999, dest.data_ptr(), hint.data_ptr(), ws.data_ptr(), st)
        assert rc == 0
        torch.cuda.synchronize()
        both = ws[32:64].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        s, cyc = both[:8], both[16:24]
        if i >= 5:
            rows.append([(int(s[k]) - int(s[0])) * 10 for k in (1, 2, 3, 4, 6, 7)])
            inner.append([(int(both[k]) - int(s[0])) * 10 for k in range(8, 16)])
            mhz.append(((int(cyc[7]) - int(cyc[0])) & 0xFFFFFFFF) / max(1, (int(s[7]) - int(s[0])) * 10) * 1e3)
    med = np.median(np.array(rows), axis=0)
    print(f'x{m}: ns since kernel entry (median of {len(rows)}): role {med[0]:.0f}  scanned {med[1]:.0f}  gathered {med[2]:.0f}  wave-selected {med[3]:.0f}  '
          f'select done {med[4]:.0f}  written {med[5]:.0f}  | wave_select entry/loaded/minmax/rounds {np.median(np.array(inner), axis=0).astype(int).tolist()}  uses={int(hint.cpu()[7])}  s_memtime ticks per us {np.median(mhz):.0f}', flush=True)
