cd $GRAFT_REPO_ROOT
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) | cfs: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) | nproc $(nproc) | load $(cat /proc/loadavg)"
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)" 
python -c "
import sys; sys.path.insert(0,'.')
import bench, os
from oracle import cpu_baseline as CB
print('quota', CB.cpu_quota(), 'cores', len(CB.one_socket_cores()))
for threads in (64, 32, 16):
    cores = CB.one_socket_cores()[:threads]
    CB.one_socket_cores = (lambda c: (lambda: c))(cores)
    r = bench.cpu_baseline_mt(65536, 10, 100000, 1000, 128, 128, 10, 5.0)
    print('CPU MT threads', threads, 'pairs/s %.0f' % r['value'], 'ms/step %.1f' % r['ms_per_step'], 'fastest', round(r['phases_ms_fastest_step']['total'],1))
"
