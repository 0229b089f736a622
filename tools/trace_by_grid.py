"""Per-(kernel, grid) view of a rocprofv3 *_kernel_trace.csv: the same kernel at different shapes, told apart by its grid.
python tools/trace_by_grid.py trace.csv [name substring] [skip first fraction, default 0.5 = warm-up launches]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[int(len(rows) * skip):]
acc = defaultdict(list)
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    if pat in name:
        acc[(name[:90], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k[0]:90s} grid {k[1]:>8s}x{k[2]:>4s}x{k[3]:>4s} n={len(v):5d} avg {sum(v)/len(v):8.1f} us  min {v2[0]:8.1f}  med {v2[len(v)//2]:8.1f}  max {v2[-1]:8.1f}")
