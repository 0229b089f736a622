"""The launches of ONE iteration from a rocprofv3 *_kernel_trace.csv, in order: start (us from the first), duration, grid, name.
python tools/trace_step.py trace.csv <iterations in the trace> [which iteration from the end, default 1]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
its = int(sys.argv[2]); back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# iterations are delimited by the Adam launches of the generator (the last kernels of an iteration): split on the longest gaps instead,
# which works for any recipe: an iteration = total / its launches, taken from the end
per = len(rows) // its
seg = rows[len(rows) - back * per: len(rows) - (back - 1) * per]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name)[:80]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  {r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:>4s}  {name}")
