"""The host side's share of the machine: how many CPUs this process may really use, and torch's intra-op pool sized to it.

Why this exists: torch sizes its OpenMP pool by the machine (128 threads on the 256-CPU hosts
of the MI355X boxes), not by the container's CPU quota (cgroup ``cpu.max``: 16 CPUs there).  One CPU-side tensor op above the
parallel grain -- the tracker's hand-back to the host has a few -- wakes the whole pool, whose threads then spin for a few
milliseconds; 128 spinning threads spend a 16-CPU quota in ~12 ms, and the kernel freezes EVERY thread of the container
until the 100 ms period ends.  Seen from the tracker: the host blocks for 15 .. 85 ms in the middle of a launch and the GPU
runs dry, a few milliseconds behind a quarter of the hand-backs (``cpu.stat``: nr_throttled 2 -> 97 over 60 hand-backs; with
8 threads: unchanged, no pause).  The tracker's own host work is index arithmetic on arrays of a megabyte: it has no use for
the pool."""
import os
import warnings

import torch

_fitted = [False]


def cpu_quota():
    """CPUs this process may use: the affinity mask, capped by the cgroup's bandwidth quota (v2 ``cpu.max``, v1
    ``cpu.cfs_quota_us / cpu.cfs_period_us``) of its own cgroup or the container root"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    paths = []
    try:
        for line in open("/proc/self/cgroup"):
            rel = line.strip().split(":", 2)[2].lstrip("/")
            paths += [os.path.join("/sys/fs/cgroup", rel), os.path.join("/sys/fs/cgroup/cpu", rel)]
    except (OSError, IndexError):
        pass
    paths += ["/sys/fs/cgroup", "/sys/fs/cgroup/cpu"]
    for base in paths:
        try:
            quota, period = open(os.path.join(base, "cpu.max")).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(quota) // int(period)))
            continue
        except (OSError, ValueError):
            pass
        try:
            quota = int(open(os.path.join(base, "cpu.cfs_quota_us")).read())
            period = int(open(os.path.join(base, "cpu.cfs_period_us")).read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def fit_host_threads(force=False):
    """once per process: torch's intra-op pool down to HALF the CPU quota (this rank's share of it: quota / LOCAL_WORLD_SIZE
    under torchrun) when it is larger than that (half: the pool's
    threads spin after every parallel region, next to the launching thread and the runtime's own threads).  A pool that
    fits is left alone; ``RAMP_HOST_THREADS=0`` leaves any pool alone, ``RAMP_HOST_THREADS=n`` sets n.  Returns the
    pool size in effect."""
    if _fitted[0] and not force:
        return torch.get_num_threads()
    _fitted[0] = True
    env = os.environ.get("RAMP_HOST_THREADS", "")          # env: the host pool (0: hands off)
    if env == "0":
        return torch.get_num_threads()
    have, quota = torch.get_num_threads(), cpu_quota()
    try:        # one process per GPU on a node (torchrun): the ranks share the container's quota
        quota = max(1, quota // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))
    except ValueError:
        pass
    want = int(env) if env else max(1, quota // 2)
    if env or have > quota:
        torch.set_num_threads(want)
        if not env:
            warnings.warn("rampvo_amd: torch's intra-op pool had %d threads, this process may use %d CPUs (affinity / cgroup quota): "
                          "pool set to %d -- a pool larger than the quota gets the whole process frozen by the kernel's "
                          "bandwidth control for tens of milliseconds at a time (RAMP_HOST_THREADS=0 leaves it alone)"
                          % (have, quota, want), stacklevel=2)
    return torch.get_num_threads()
