#!/bin/bash
# tools/stall_cgroup.sh: is the pause CPU-bandwidth throttling of the container?  cpu.max / cpu.stat of this cgroup around runs of
# tools/stall_hunt.py with the host thread pools at their default size, at 1 and at 8 threads
show() { for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us /sys/fs/cgroup/cpu/cpu.stat; do [ -r $f ] && { echo "-- $f"; cat $f | tr '\n' ' '; echo; }; done; }
echo "nproc=$(nproc) online=$(cat /sys/devices/system/cpu/online)"; python -c "import os, torch; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads(), 'interop', torch.get_num_interop_threads())"
cat /proc/self/cgroup
show
for e in "X=1" "OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1" "OMP_NUM_THREADS=8" "OMP_WAIT_POLICY=passive GOMP_SPINCOUNT=0"; do
  echo "== $e"; env $e python tools/stall_hunt.py ${1:-60} 30 plain settle 2>&1 | grep -E "cycles with"; show | grep -A1 "cpu.stat"
done
