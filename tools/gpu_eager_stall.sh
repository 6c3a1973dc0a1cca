#!/bin/bash
# Evidence for the periodic host stall of eagerly launched test images (profiles/r03_eager_stall.txt): per-image wall
# times with the default intra-op pool, the cgroup CPU statistics before / after, the HIP-API accounting per image (the
# stall is a gap BETWEEN two HIP calls, no call is slow), and the same loop with the pool capped at the CPU quota.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/eager_stall.txt; cd /tmp; export TMPDIR=/tmp
stat() { grep -E "nr_periods|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
{
echo "host threads: $(nproc)   /sys/fs/cgroup/cpu.max: $(cat /sys/fs/cgroup/cpu.max)"
echo; echo "== default intra-op pool"; echo "cpu.stat before: $(stat)"
timeout 300 python $R/tools/stall_probe.py 16 2>&1 | tail -2
echo "cpu.stat after:  $(stat)"
echo; echo "== RESPECT_QUOTA=1 (detectron_pytorch_amd.hostcpu.respect_cpu_quota)"; echo "cpu.stat before: $(stat)"
RESPECT_QUOTA=1 timeout 300 python $R/tools/stall_probe.py 16 2>&1 | tail -3
echo "cpu.stat after:  $(stat)"
echo; echo "== OMP_NUM_THREADS=1"
OMP_NUM_THREADS=1 timeout 300 python $R/tools/stall_probe.py 16 2>&1 | tail -2
echo; echo "== GC=report (Python's collector timed: not the cause)"
GC=report timeout 300 python $R/tools/stall_probe.py 12 2>&1 | tail -3
echo; echo "== HIP API accounting per image, default pool (rocprofv3 --hip-trace --kernel-trace; tools/slow_launches.py)"
timeout 400 rocprofv3 --hip-trace --kernel-trace -d $R/gpurun_out/stall_hip -o s -f csv -- python $R/tools/stall_probe.py 12 > $R/gpurun_out/stall_hip.log 2>&1
grep "ms per image" $R/gpurun_out/stall_hip.log
python $R/tools/slow_launches.py $R/gpurun_out/stall_hip 15 10000 | cut -c1-420 | tail -14
rm -rf $R/gpurun_out/stall_hip
} > $O 2>&1
cat $O | cut -c1-260
