"""Developer aid (a -DPPQHIP_QH_TIMING build): s_memrealtime stamps of workgroup 0 through the EXACT passes of quantile_hot_select_kernel
(the hint is zeroed before every call).  Per level: start, chunks walked, counts flushed, level complete, bins picked."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppq_amd._lib import lib
dev = torch.device('cuda'); g = torch.Generator(device=dev).manual_seed(0); st = torch.cuda.current_stream().cuda_stream
for m in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '1,32').split(',')]:
    n = m * 512 * 56 * 56
    x = torch.randn(n, device=dev, generator=g)
    ws = torch.zeros(int(lib.ppqhip_quantile_workspace_bytes(n)) // 4 + 16, dtype=torch.int32, device=dev)
    hint = torch.zeros(8, dtype=torch.int32, device=dev); dest = torch.zeros(2, device=dev)
    rows = []
    for i in range(20):
        hint.zero_()
        assert lib.ppqhip_quantile_t(x.data_ptr(), n, 0.9999, dest.data_ptr(), hint.data_ptr(), ws.data_ptr(), st) == 0
        torch.cuda.synchronize()
        s = ws[32:64].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        if i >= 3: rows.append([(int(s[k]) - int(s[0])) * 10 for k in [1, 6] + list(range(16, 28)) + [7]])
    med = np.median(np.array(rows), axis=0).astype(int).tolist()
    print(f'x{m}: role {med[0]} decided {med[1]} | ' + ' | '.join(f'L{l}: start {med[2 + 4 * l]} walked {med[3 + 4 * l]} drained {med[4 + 4 * l]} complete {med[5 + 4 * l]}' for l in range(3)) + f' | written {med[14]}  [ns since kernel entry]')
