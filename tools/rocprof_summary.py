#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (``--kernel-trace --stats``) into a per-kernel table.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/r01_x_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = db.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print('%-86s %7s %12s %11s %11s %11s %6s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', '%'))
    for n, c, tot, avg, mn, mx in rows:
        print('%-86s %7d %12d %11.0f %11d %11d %6.2f' % (n[:86], c, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == '__main__':
    main(sys.argv[1])
