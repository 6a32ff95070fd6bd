#!/usr/bin/env python3
"""Per-kernel means of the counters in rocprofv3 `*_results.db` files (rocpd SQLite, `--pmc` passes), plus dispatch
count and mean duration: the successor of pmc_summary.py for the database output format."""
import glob
import sqlite3
import sys


def main(dirs):
    for d in dirs:
        for path in sorted(glob.glob(d + "/**/*_results.db", recursive=True)):
            db = sqlite3.connect(path)
            print("==", path)
            dur = {k: (n, a) for k, n, a in db.execute("select name, count(*), avg(end - start) from kernels group by name")}
            rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                              "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
            last = None
            for k, c, v, n in rows:
                if k != last:
                    nd, ad = dur.get(k, (0, 0.0))
                    print(f"{k[:150]}\n    dispatches={nd} avg_us={ad / 1e3:.1f}")
                    last = k
                print(f"    {c:28s} mean={v:.6g}  (n={n})")


if __name__ == "__main__":
    main(sys.argv[1:])
