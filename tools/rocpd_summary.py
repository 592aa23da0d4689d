#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a text table:
per kernel: calls, total/avg/min/max duration (us), grid, VGPRs, LDS.
    python tools/rocpd_summary.py results.db [> profiles/rNN_xxx.txt]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\[clone .*\]', '', name)
    name = name.replace('sert::', '').replace('void ', '')
    return name if len(name) <= 90 else name[:87] + '...'


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(grid_y), max(grid_z), max(workgroup_x), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print('%-90s %7s %12s %10s %10s %10s %6s  %-18s %5s %5s %5s %7s' % (
        'kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%', 'grid(threads)', 'vgpr',
        'agpr', 'sgpr', 'lds'))
    for r in rows:
        print('%-90s %7d %12.1f %10.2f %10.2f %10.2f %6.2f  %-18s %5d %5d %5d %7d' % (
            short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot,
            '%dx%dx%d/%d' % (r[6], r[7], r[8], r[9]), r[10] or 0, r[11] or 0, r[12] or 0, r[13] or 0))
    print('total kernel time: %.1f us over %d dispatches' % (tot / 1e3, sum(r[1] for r in rows)))


if __name__ == '__main__':
    main(sys.argv[1])
