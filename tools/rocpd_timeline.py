#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 rocpd database (kernel trace): every dispatch
between two consecutive launches of the step's first kernel, with its start offset, duration,
queue and the idle gap of its queue before it.
    python tools/rocpd_timeline.py results.db [first_kernel_substring] [which_step_from_the_end]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\[clone .*\]', '', name).replace('sert::', '').replace('void ', '')
    return name.split('(')[0][:70]


def main(path, first='vs_gather_mean', back=3):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    q = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else 'tid')
    rows = db.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    a, b = starts[-back - 1], starts[-back]
    t0 = rows[a][1]
    last_end = {}
    print('%-70s %6s %10s %9s %9s' % ('kernel', 'queue', 'start_us', 'dur_us', 'gap_us'))
    for name, s, e, qid in rows[a:b]:
        gap = (s - last_end[qid]) / 1e3 if qid in last_end else float('nan')
        print('%-70s %6s %10.1f %9.1f %9.1f' % (short(name), qid, (s - t0) / 1e3, (e - s) / 1e3, gap))
        last_end[qid] = e
    print('step span: %.1f us   sum of durations: %.1f us' % (
        (rows[b][1] - t0) / 1e3, sum(r[2] - r[1] for r in rows[a:b]) / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], *(sys.argv[2:3] or ['vs_gather_mean']), *[int(x) for x in sys.argv[3:4]])
