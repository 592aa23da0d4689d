#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE are collected in SEPARATE runs: they do not fit one TCC pass).

    python tools/rocpd_pmc.py fetch_results.db write_results.db [--json out.json]

Units and corrections (MI355X_MICROARCH.md, section HBM): the counters are in
KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
streaming read, so the read side is DOUBLED here ("fetch_bytes_corrected");
WRITE_SIZE is taken as reported (uncalibrated per the guide).
"""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\[clone .*\]', '', name).replace('sert::', '').replace('void ', '')
    return name.split('(')[0]


def per_kernel(path, counter):
    """Keyed by 'kernel @grid': the same kernel launched at different sizes (e.g. the
    optimiser over a 12.8 M-element table and over a 128-element bias) stays apart.  Launches of
    one kernel with ONE grid size but clearly different durations (the C4 step runs adam_l2 over
    the 150 M-element word table and the 30 M-element entity table with the same 4096 workgroups)
    are split into duration clusters, keyed 'kernel @grid #k' in ascending duration."""
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, grid_size, value, duration from counters_collection "
                      "where counter_name = ?", (counter,)).fetchall()
    groups = {}
    for name, grid, value, dur in rows:
        groups.setdefault((short(name), grid), []).append((dur, value))
    out = {}
    for (name, grid), items in groups.items():
        items.sort()
        clusters, cur = [], [items[0]]
        for it in items[1:]:
            if it[0] > 1.5 * cur[-1][0] and it[0] - cur[-1][0] > 5000:   # (ns) a gap of > 50 % and > 5 us
                clusters.append(cur)
                cur = []
            cur.append(it)
        clusters.append(cur)
        for k, cl in enumerate(clusters):
            key = '%s @%d' % (name, grid) + ('' if len(clusters) == 1 else ' #%d' % k)
            out[key] = dict(calls=len(cl), kib=sum(v for _, v in cl) / len(cl), us=sum(d for d, _ in cl) / len(cl) / 1e3)
    return out


def main(argv):
    fetch = per_kernel(argv[1], 'FETCH_SIZE')
    write = per_kernel(argv[2], 'WRITE_SIZE')
    out = {}
    print('%-60s %6s %12s %14s %12s %12s %10s' % ('kernel', 'calls', 'fetch_MB_raw', 'fetch_MB_x2',
                                                   'write_MB', 'hbm_MB', 'avg_us'))
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, {}).get('kib', 0))):
        f = fetch.get(k, {}).get('kib', 0.0) * 1024
        w = write.get(k, {}).get('kib', 0.0) * 1024
        us = fetch.get(k, write.get(k, {})).get('us', 0.0)
        out[k] = dict(fetch_bytes_raw=f, fetch_bytes_corrected=2 * f, write_bytes=w,
                      hbm_bytes=2 * f + w, avg_us_profiled=us,
                      calls=fetch.get(k, write.get(k, {})).get('calls', 0))
        print('%-60s %6d %12.2f %14.2f %12.2f %12.2f %10.2f' % (
            k[:60], out[k]['calls'], f / 1e6, 2 * f / 1e6, w / 1e6, (2 * f + w) / 1e6, us))
    if '--json' in argv:
        with open(argv[argv.index('--json') + 1], 'w') as fh:
            json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == '__main__':
    main(sys.argv)
