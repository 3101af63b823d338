#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output) as a kernel-stats table.
usage: python tools/rocpd_summary.py <results.db> [more.db ...]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f"# {path}")
        print("name,calls,total_us,avg_us,pct")
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            short = name.split("(")[0]
            print(f'"{short}",{calls},{total / 1e0:.3f},{avg:.3f},{pct:.2f}')


if __name__ == "__main__":
    main()
