#!/usr/bin/env python
"""Print per-kernel averages of every PMC counter in a rocpd database."""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, grid_size, counter_name, avg(value), count(*), avg(duration) from "
                  "counters_collection group by kernel_name, grid_size, counter_name").fetchall()
out = {}
for name, grid, cn, v, n, d in rows:
    key = (re.sub(r'\[clone .*\]', '', name).replace('sert::', '').replace('void ', '').split('(')[0], grid)
    out.setdefault(key, {})[cn] = v
    out[key]['_us'] = d / 1e3
    out[key]['_n'] = n
for key in sorted(out, key=lambda k: -out[k]['_us']):
    c = out[key]
    print('%s @%d  n=%d  %.1f us' % (key[0], key[1], c['_n'], c['_us']))
    for k in sorted(c):
        if not k.startswith('_'):
            print('    %-32s %16.1f' % (k, c[k]))
