"""Host-side CPU budget of the process: the cgroup CPU quota of the container and PyTorch's intra-op thread pool.

Why this exists (measured, profiles/r03_eager_stall.txt): on a 256-thread host whose container has a 16-CPU quota
(/sys/fs/cgroup/cpu.max = "1600000 100000"), PyTorch sizes its intra-op pool from the 256 visible threads.  Every small
CPU-side op of an eagerly launched test image (result formats, im_info arithmetic) wakes the pool, the OpenMP workers
spin after the parallel region, the cgroup burns its quota within a 100 ms CFS period and the kernel throttles EVERY
thread of the process -- the launching thread included -- until the period ends: every third or fourth image of
rcnn.inference.im_detect_all_results took 70-80 ms instead of 10 ms (cpu.stat: nr_throttled grows; no HIP call is
slow, the time is a gap between two HIP calls).  With the pool capped to the quota the images take 10.2 ms flat.
The hipGraph entry points were never affected (no CPU ops in the replayed region)."""
import os
import warnings

import torch


def cpu_quota():
    """CPUs the cgroup of this process may use per scheduling period (float), or None when unlimited / unknown."""
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
        return None
    except (OSError, ValueError):
        pass
    try:  # cgroup v1
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return float(quota) / period if quota > 0 and period > 0 else None
    except (OSError, ValueError):
        return None


def respect_cpu_quota(share=0.25, processes=None):
    """Cap torch's intra-op pool at `share` of the cgroup quota divided by the processes that share it (default: the
    LOCAL_WORLD_SIZE torchrun exports, else 1); at least 1 thread -- the launching thread, the HSA runtime threads and
    MIOpen's need the rest.  No-op without a quota or when OMP_NUM_THREADS is set.  Returns the thread count."""
    quota = cpu_quota()
    if processes is None:
        try:
            processes = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        except ValueError:
            processes = 1
    if quota is not None and "OMP_NUM_THREADS" not in os.environ:
        want = max(1, int(quota * share / processes))
        if torch.get_num_threads() > want:
            torch.set_num_threads(want)
    return torch.get_num_threads()


def warn_if_oversubscribed():
    """One warning when the intra-op pool is larger than the cgroup quota (see the module docstring)."""
    quota = cpu_quota()
    if quota is not None and torch.get_num_threads() > quota:
        warnings.warn("torch uses %d intra-op threads but the container's CPU quota is %.0f CPUs: eager launches will be "
                      "throttled periodically; call detectron_pytorch_amd.hostcpu.respect_cpu_quota() or set "
                      "OMP_NUM_THREADS" % (torch.get_num_threads(), quota), RuntimeWarning, stacklevel=2)
