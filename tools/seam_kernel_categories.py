"""Sums a rocprofv3 kernel-trace CSV per category (this library / torch reductions / MIOpen + rocBLAS / other)."""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
passes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cat = collections.defaultdict(lambda: [0, 0.0])
top = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'ppqhip' in n: c = 'this library (ppqhip::*)'
    elif 'reduce_kernel' in n or 'MinOps' in n or 'MaxOps' in n or 'min_' in n.lower() and 'at::native' in n: c = "torch reductions (the observers' value.min() / value.max() ...)"
    elif any(k in n for k in ('miopen', 'MIOpen', 'Cijk', 'igemm', 'naive_conv', 'gridwise', 'conv', 'Conv', 'gemm', 'SubTensorOp', 'batchnorm', 'BatchNorm', 'pooling', 'Pooling')): c = 'MIOpen / rocBLAS (the network)'
    else: c = 'other torch kernels (elementwise, copies, cat ...)'
    cat[c][0] += 1; cat[c][1] += d
    k = n.replace('void ', '').split('(')[0][:90]
    top[k][0] += 1; top[k][1] += d
tot = sum(v[1] for v in cat.values())
print(f'kernel time per pass ({passes:g} passes in the trace): {tot / passes / 1e3:.2f} ms')
for c, (k, d) in sorted(cat.items(), key=lambda kv: -kv[1][1]): print(f'  {d / passes / 1e3:9.2f} ms  {100 * d / tot:5.1f} %  {k / passes:8.0f} launches  {c}')
print('  top kernels:')
for k, (cnt, d) in sorted(top.items(), key=lambda kv: -kv[1][1])[:12]: print(f'  {d / passes / 1e3:9.2f} ms  {cnt / passes:8.0f}  {k}')
