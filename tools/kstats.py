"""Per-iteration view of a rocprofv3 kernel_stats.csv:  python tools/kstats.py file.csv iterations [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('kernel time / iteration %.2f ms, launches / iteration %.0f' % (tot / 1e6 / it, calls / it))
for r in rows[:top]:
    print('%6.1f/it %7.1f us/it avg %7.2f  %s' % (int(r['Calls']) / it, float(r['TotalDurationNs']) / 1e3 / it, float(r['AverageNs']) / 1e3, r['Name'][:100]))
