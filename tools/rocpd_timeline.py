#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 kernel trace (rocpd sqlite):
every dispatch between two consecutive launches of the anchor kernel, with its
start/end relative to the first, the queue it ran on and the idle gap before it.

    python tools/rocpd_timeline.py results.db [anchor-substring] [which-step]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\[clone .*\]', '', name).replace('sert::', '').replace('void ', '')
    return name.split('(')[0][:60]


def main(path, anchor='vs_gather_mean', which=60):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    rows = db.execute("select name, start, end, %s from kernels order by start" % (qcol or '0')).fetchall()
    anchors = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(anchors) < which + 2:
        which = len(anchors) // 2
    i0, i1 = anchors[which], anchors[which + 1]
    t0 = rows[i0][1]
    print('step window: %.1f us, %d dispatches' % ((rows[i1][1] - t0) / 1e3, i1 - i0))
    last_end = t0
    busy = 0.0
    for name, st, en, q in rows[i0:i1]:
        gap = (st - last_end) / 1e3
        print('%8.1f %8.1f %7.1f  q=%-4s gap=%6.1f  %s' % ((st - t0) / 1e3, (en - t0) / 1e3,
                                                         (en - st) / 1e3, q, gap, short(name)))
        last_end = max(last_end, en)
    # union of busy intervals
    iv = sorted((r[1], r[2]) for r in rows[i0:i1])
    cur_s, cur_e = iv[0]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print('GPU busy (union of kernels): %.1f us of %.1f' % (busy / 1e3, (rows[i1][1] - t0) / 1e3))


if __name__ == '__main__':
    a = sys.argv[1:]
    main(a[0], a[1] if len(a) > 1 else 'vs_gather_mean', int(a[2]) if len(a) > 2 else 60)
