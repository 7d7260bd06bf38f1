"""Largest kernels and largest inter-kernel gaps in a rocprofv3 kernel-trace CSV (tail = timed region)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t_end = int(rows[-1]['End_Timestamp'])
rows = [r for r in rows if int(r['Start_Timestamp']) > t_end - int(float(sys.argv[2]) * 1e6)]   # last N ms
print('kernels in window', len(rows))
big = sorted(rows, key=lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']), reverse=True)[:6]
for r in big:
    print('  long %8.1f us  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:70]))
gaps = []
for a, b in zip(rows, rows[1:]):
    gaps.append((int(b['Start_Timestamp']) - int(a['End_Timestamp']), a['Kernel_Name'][:50], b['Kernel_Name'][:50]))
for g in sorted(gaps, reverse=True)[:6]:
    print('  gap  %8.1f us  after %s  before %s' % (g[0] / 1e3, g[1], g[2]))
