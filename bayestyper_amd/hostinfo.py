"""What the host gives this process (bench / tools harness only): logical CPUs, the scheduler affinity mask and the cgroup CPU quota.
A container on a 256-thread box may be limited to a few cores' worth of time (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`): threads beyond the quota
are throttled, so a CPU baseline's "cores" is the quota, not os.cpu_count()."""
import math
import os


def host_facts():
    f = {"cpu_count": os.cpu_count() or 1, "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)}
    quota = None
    try:   # cgroup v2
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        f["cgroup_cpu_max"] = " ".join(q)
        if q and q[0] != "max":
            quota = float(q[0]) / float(q[1])
    except (OSError, ValueError, IndexError):
        pass
    if quota is None:
        try:   # cgroup v1
            qu = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pe = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            f["cgroup_cfs_quota_us"], f["cgroup_cfs_period_us"] = qu, pe
            if qu > 0 and pe > 0:
                quota = qu / pe
        except (OSError, ValueError):
            pass
    f["cgroup_cores"] = quota
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                f["model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    f["effective_cores"] = min(f["affinity"], quota) if quota else float(f["affinity"])
    return f


def baseline_threads(facts):
    """threads for an all-cores CPU leg: two per effective core (a quota is CPU time, not a core set: a few more runnable threads than cores keep it used
    while one is descheduled; ten times more are throttled — measured on the round-6 box: 16-core quota, 32 threads 16.7x one thread, 256 threads 9.0x)"""
    return int(max(1, min(facts["affinity"], math.ceil(2 * facts["effective_cores"]))))
