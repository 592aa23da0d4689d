# which HIP API calls does the host make between the last kernel of a run-ahead backward and the word-table update of the next call?
mkdir -p gpurun_out/r05e; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ht
rocprofv3 --hip-trace --kernel-trace -d /tmp/ht -o t -- python $R/bench.py --profile-inner --num-batches 8 --batch ${BATCH:-8192} --steps 12 --warmup 3 > /dev/null 2>&1
DB=$(find /tmp/ht -name '*.db' | head -1)
python - "$DB" <<'P' > $R/gpurun_out/r05e/hip_trace_${TAG:-8192}.txt
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print('tables/views:', tabs)
def cols(t): return [r[1] for r in db.execute('pragma table_info(%s)' % t)]
for t in tabs:
    if 'region' in t.lower() or 'api' in t.lower():
        print(t, cols(t))
ks = db.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(ks) if 'vs_gather_mean' in r[0]]
a, b = ks[starts[-4]][1], ks[starts[-2]][1]
ev = [(s, 'K ' + re.sub(r'\(.*', '', n.replace('sert::', '').replace('void ', ''))[:60] + '  dur %.1f' % ((e - s) / 1e3)) for n, s, e in ks if a <= s < b]
for t in ('regions', 'regions_and_samples'):
    if t in tabs:
        c = cols(t)
        if 'name' in c and 'start' in c:
            for n, s, e in db.execute("select name, start, end from %s where start >= ? and start < ? order by start" % t, (a - 300000, b)):
                ev.append((s, 'H ' + str(n) + '  dur %.1f' % ((e - s) / 1e3)))
            break
ev.sort()
for s, txt in ev:
    print('%10.1f  %s' % ((s - a) / 1e3, txt))
P
tail -150 $R/gpurun_out/r05e/hip_trace_${TAG:-8192}.txt
