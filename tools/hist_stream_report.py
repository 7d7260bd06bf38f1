"""Splits the rocprofv3 kernel trace of tools/hist_stream_bench.py at its separator launches and prints, per case, the median device
duration of every kernel of the case and their sum (a call that launches two kernels = the sum of their medians)."""
import collections
import csv
import sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
names = sys.argv[2].split('|')
segs, cur = [], []
for r in rows:
    n = r['Kernel_Name']
    if 'ppqhip' not in n: continue
    if 'hist_small_kernel' in n and int(r['Grid_Size_X']) == 512 and cur:        # the separator: one workgroup
        segs.append(cur); cur = []
        continue
    cur.append(r)
for name, seg in zip(names, segs):
    agg = collections.defaultdict(list)
    for r in seg: agg[r['Kernel_Name'].replace('void ', '').replace('ppqhip::', '').split('(')[0] + ' grid=' + str(int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    tot = 0.0; parts = []
    for k, v in agg.items():
        v.sort(); med = v[len(v) // 2] / 1e3
        if len(v) >= 10: tot += med; parts.append(f'{k} {med:.2f}')
    print(f'{name:28s} {tot:7.2f} us   ' + ' + '.join(parts))
