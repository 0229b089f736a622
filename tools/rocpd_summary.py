"""Summarise a rocprofv3 `--kernel-trace --stats` result (rocpd SQLite database) as text:
per-kernel calls / total / average duration / share, like the CSV stats table.

    python tools/rocpd_summary.py gpurun_out/prof1/r01_results.db > profiles/r01_bench_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) <= 150 else name[:147] + "..."


def main(path: str, top: int = 60) -> None:
    con = sqlite3.connect(path)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    print(f"# source: {path}")
    print(f"# kernels: {len(rows)}   total device time: {total / 1e3:.3f} ms   (durations in us)")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  name")
    for name, calls, tot, avg, pct in rows[:top]:
        print(f"{calls:7d} {tot:12.1f} {avg:10.2f} {pct:6.2f}  {short(name)}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
