"""Per-case device-duration medians from a rocprofv3 kernel trace of tools/floor_table.py.

  python tools/floor_report.py <kernel_trace.csv> gpurun_out/floor_manifest_<tag>.json

The trace is cut at every `floor_marker_kernel` launch (grid size / 64 = case id); inside a case the first `warmup` launches of
each kernel are dropped and the rest reported as median / p10 / min of (End - Start) per kernel name, plus the case's SUM over its
kernels when a case launches more than one kernel per call (one-shot histogram + reduce, quantile sequences).
"""
import collections
import csv
import json
import sys


def short(name):
    return name.replace('void ', '').replace('ppqhip::', '').split('(')[0]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    man = json.load(open(sys.argv[2]))
    cases = {int(k): v for k, v in man['cases'].items()}
    warm, iters = man['warmup'], man['iters']
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    cur, segs = None, []                          # one segment per marker launch: (case id, {kernel key: [durations]})
    for r in rows:
        nm = r['Kernel_Name']
        if 'floor_marker_kernel' in nm:
            cid = int(r['Grid_Size_X']) // 64
            cur = None
            if cid in cases:
                cur = collections.OrderedDict()
                segs.append((cid, cur))
            continue
        if cur is None: continue
        key = (short(nm), int(r['Grid_Size_X']), int(r['Workgroup_Size_X']))
        cur.setdefault(key, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    seg = collections.OrderedDict()               # pooled over rounds, the warm-up launches of every segment dropped
    for cid, kernels in segs:
        pooled = seg.setdefault(cid, collections.OrderedDict())
        for key, v in kernels.items():
            calls = len(v) / float(iters + warm)
            pooled.setdefault(key, []).extend(v[int(round(warm * calls)):])
    l2l = {n: u for n, u in man.get('launch_to_launch_us', [])}
    print(f'# shape {man["shape"]}  library {man.get("library")}  {iters} timed launches per case (first {warm} dropped); device durations from rocprofv3 --kernel-trace [us]')
    print(f'# {"case":88s} {"kernel":44s} {"grid":>6s} {"wg":>5s} {"n":>4s} {"median":>7s} {"p10":>7s} {"min":>7s} {"launch-to-launch":>17s}')
    for cid, kernels in seg.items():
        name = cases[cid]
        per_call = 0.0
        for (k, g, w), v in kernels.items():
            calls = len(v) / float(iters * man.get('rounds', 1))
            v = sorted(v)
            if not v: continue
            med = v[len(v) // 2] / 1e3
            per_call += med * max(1, round(calls)) if calls >= 0.5 else 0.0
            print(f'  {name:88s} {k[:44]:44s} {g // max(w, 1):6d} {w:5d} {len(v):4d} {med:7.2f} {v[len(v) // 10] / 1e3:7.2f} {v[0] / 1e3:7.2f} {l2l.get(name, float("nan")):17.2f}')
        if len(kernels) > 1:
            print(f'  {name:88s} {"= SUM of the medians of its kernels":44s} {"":6s} {"":5s} {"":4s} {per_call:7.2f}')


if __name__ == '__main__':
    main()
