"""Launch-by-launch view of one kernel in a rocprofv3 *_kernel_trace.csv: duration, gap to and name of the launch before it.
python tools/trace_context.py trace.csv <name substring> <Grid_Size_X> [max rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat, gx = sys.argv[2], sys.argv[3]
lim = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]
n = 0
for i, r in enumerate(rows):
    if pat in r['Kernel_Name'] and r['Grid_Size_X'] == gx and i > 1:
        p, q = rows[i - 1], rows[i + 1] if i + 1 < len(rows) else None
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        gap = (int(r['Start_Timestamp']) - int(p['End_Timestamp'])) / 1e3
        pd = (int(p['End_Timestamp']) - int(p['Start_Timestamp'])) / 1e3
        short = lambda s: re.sub(r'\(anonymous namespace\)::', '', s)[:70]
        print(f"{d:8.1f} us  gap {gap:6.1f}  prev {pd:7.1f} us {short(p['Kernel_Name'])}")
        n += 1
        if n >= lim:
            break
