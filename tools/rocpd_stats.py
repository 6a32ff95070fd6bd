#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 `*_results.db` (rocpd SQLite output of
`rocprofv3 --kernel-trace --stats`), written as CSV: what `profiles/*_kernel_stats.csv` hold."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                      f"from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["name,calls,total_ms,avg_us,min_us,max_us,percent"]
    for n, c, t, a, mn, mx in rows:
        n = n.replace('"', "'")
        lines.append(f'"{n[:160]}",{c},{t / 1e6:.3f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * t / total:.2f}')
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
