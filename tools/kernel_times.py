"""Median device duration per (kernel, grid) from a rocprofv3 kernel-trace CSV."""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'ppqhip' not in n: continue
    if len(sys.argv) > 2 and sys.argv[2] not in n: continue
    k = n.replace('void ', '').replace('ppqhip::', '').split('(')[0]
    agg[(k, int(r['Grid_Size_X']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for (k, g), v in sorted(agg.items()):
    v.sort()
    print(f'{k:52s} threads={g:9d} n={len(v):4d} median={v[len(v)//2]/1e3:8.2f}us min={v[0]/1e3:8.2f}us')
