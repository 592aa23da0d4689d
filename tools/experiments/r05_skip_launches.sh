# dense_update_skip launch by launch (a full pass every kLazyK updates, the others read what the batch or the next one touches)
mkdir -p gpurun_out/r05h; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128" "c2_8192 --batch 8192" "w3c --model loglinear --batch 1024 --window 8 --entities 715 --dim 300" "c4 --vocab 500000 --entities 100000 --dim 300"; do
  set -- $cfg; name=$1; shift
  rm -rf /tmp/sk_$name
  rocprofv3 --kernel-trace -d /tmp/sk_$name -o t -- python $R/bench.py --profile-inner --num-batches 8 "$@" --steps 24 --warmup 8 > /dev/null 2>&1
  DB=$(find /tmp/sk_$name -name '*.db' | head -1)
  python - $DB $name <<'P' | tee -a $R/gpurun_out/r05h/skip_launches.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
d = [(e - s) / 1e3 for n, s, e in rows if 'dense_update_skip' in n or 'dense_update_lazy' in n]
print(sys.argv[2], 'last 16 launches, us:', ' '.join('%.0f' % x for x in d[-16:]))
P
done
