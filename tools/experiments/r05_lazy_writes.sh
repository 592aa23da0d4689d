# WRITE_SIZE / FETCH_SIZE of every launch of the lazy word-table update at C4, in launch order: what does a step really write?
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in WRITE_SIZE FETCH_SIZE; do
rm -rf /tmp/lw_$c
rocprofv3 --kernel-trace --pmc $c -d /tmp/lw_$c -o w -- python $R/bench.py --profile-inner --num-batches 8 --vocab 500000 --entities 100000 --dim 300 --steps 12 --warmup 3 > /dev/null 2>&1
DB=$(find /tmp/lw_$c -name '*.db' | head -1)
python - "$DB" $c <<'P'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute('pragma table_info(counters_collection)')]
rows = db.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%dense_update_lazy%' order by dispatch_id").fetchall()
agg = {}
for did, kn, cn, v, d in rows:
    a = agg.setdefault(did, [0.0, d]); a[0] += v
print(sys.argv[2], 'per launch (raw counter units: KiB), launch order:')
print(' '.join('%.0f' % (a[0] / 1024.0) for _, a in sorted(agg.items())), '(MiB)')
P
done
