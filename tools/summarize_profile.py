"""Condense a rocprofv3 output directory (trace/ fetch/ write/ sub-dirs of CSVs) into the small
summaries kept under profiles/: kernel stats (top N + every ppqhip kernel) and PMC bytes per launch."""
import collections
import csv
import json
import os
import sys


def short(name):
    n = name.replace('void ', '').replace('ppqhip::', '')
    return n.split('(')[0][:90]


def main(src, dst_prefix, top=25):
    rows = list(csv.DictReader(open(os.path.join(src, 'trace', [f for f in os.listdir(os.path.join(src, 'trace')) if f.endswith('kernel_stats.csv')][0]))))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    with open(dst_prefix + '_kernel_stats.csv', 'w') as f:
        f.write('kernel,calls,total_ms,avg_us,min_us,max_us,percent\n')
        for i, r in enumerate(rows):
            if i < top or 'ppqhip' in r['Name']:
                f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},"
                        f"{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f},{100*float(r['TotalDurationNs'])/total:.2f}\n")
    pmc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for sub in ('fetch', 'write'):
        d = os.path.join(src, sub)
        if not os.path.isdir(d): continue
        fn = [f for f in os.listdir(d) if f.endswith('counter_collection.csv')][0]
        for r in csv.DictReader(open(os.path.join(d, fn))):
            if 'ppqhip' not in r['Kernel_Name']: continue
            e = pmc[short(r['Kernel_Name'])][r['Counter_Name']]
            e[0] += 1; e[1] += float(r['Counter_Value'])
    out = {}
    for k, cs in pmc.items():
        o = {}
        for c, (n, tot) in cs.items():
            o[c + '_KB_per_launch'] = round(tot / n, 1); o['launches'] = n
        # MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
        # coalesced streams -> double it; WRITE_SIZE is taken as reported (uncalibrated).
        if 'FETCH_SIZE_KB_per_launch' in o:
            o['hbm_read_bytes_per_launch_corrected'] = round(o['FETCH_SIZE_KB_per_launch'] * 1024 * 2)
        if 'WRITE_SIZE_KB_per_launch' in o:
            o['hbm_write_bytes_per_launch'] = round(o['WRITE_SIZE_KB_per_launch'] * 1024)
        out[k] = o
    json.dump(out, open(dst_prefix + '_pmc_bytes.json', 'w'), indent=1, sort_keys=True)
    print(open(dst_prefix + '_kernel_stats.csv').read()[:3000])
    print(json.dumps(out, indent=1)[:2500])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
