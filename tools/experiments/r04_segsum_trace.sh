# per-(kernel, grid) durations and fabric bytes of the word-gradient tree, ungrouped vs row-grouped level 0
mkdir -p gpurun_out/r04f; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in 1 8; do
  rm -rf /tmp/tr_$g /tmp/pf_$g /tmp/pw_$g
  SERT_SEG_GROUPS=$g rocprofv3 --kernel-trace -d /tmp/tr_$g -o t -- python $R/bench.py --profile-inner --steps 30 --warmup 5 > /dev/null 2>&1
  SERT_SEG_GROUPS=$g rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$g -o t -- python $R/bench.py --profile-inner --steps 12 --warmup 3 > /dev/null 2>&1
  SERT_SEG_GROUPS=$g rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$g -o t -- python $R/bench.py --profile-inner --steps 12 --warmup 3 > /dev/null 2>&1
  python - <<PY > $R/gpurun_out/r04f/segsum_g$g.txt
import sqlite3, glob, re, sys
sys.path.insert(0, '$R/tools')
import rocpd_pmc
db = sqlite3.connect(glob.glob('/tmp/tr_$g/**/*.db', recursive=True)[0])
print('SERT_SEG_GROUPS=$g  kernel trace (30 steps)')
for name, grid, n, avg in db.execute("select name, grid_x, count(*), avg(duration) from kernels where name like '%segsum%' or name like '%adam_l2%' group by name, grid_x order by avg(duration) desc"):
    print('%-60s grid %8d calls %4d avg %8.2f us' % (re.sub(r'\[clone.*', '', name)[:60], grid, n, avg / 1e3))
f = rocpd_pmc.per_kernel(glob.glob('/tmp/pf_$g/**/*.db', recursive=True)[0], 'FETCH_SIZE')
w = rocpd_pmc.per_kernel(glob.glob('/tmp/pw_$g/**/*.db', recursive=True)[0], 'WRITE_SIZE')
for k in sorted(f):
    if 'segsum' in k or 'adam_l2' in k:
        print('%-70s fetch x2 %8.1f MB  write %8.1f MB  %7.2f us' % (k[:70], 2 * f[k]['kib'] * 1024 / 1e6, w.get(k, {}).get('kib', 0) * 1024 / 1e6, f[k]['us']))
PY
  cat $R/gpurun_out/r04f/segsum_g$g.txt
done
