"""bayestyper_amd/hostinfo.py: what bench.py's cpu_baseline and tools/e2e_c2.sh size their thread counts with (round 6: a container's CPU quota, not
os.cpu_count(), is the number of cores a CPU leg can use)."""
import os

from bayestyper_amd import hostinfo


def test_host_facts_are_consistent():
    f = hostinfo.host_facts()
    assert 1 <= f["affinity"] <= f["cpu_count"]
    assert 0 < f["effective_cores"] <= f["affinity"]
    if f["cgroup_cores"] is not None:
        assert f["effective_cores"] == min(f["affinity"], f["cgroup_cores"])
    t = hostinfo.baseline_threads(f)
    assert 1 <= t <= f["affinity"] and t >= min(f["affinity"], f["effective_cores"])


def test_baseline_threads_follow_the_quota_not_the_logical_cpus():
    # the round-5 GPU boxes: 256 logical CPUs, cgroup quota 16 cores -> 32 threads (two per core of the quota), not 256
    assert hostinfo.baseline_threads({"affinity": 256, "effective_cores": 16.0}) == 32
    assert hostinfo.baseline_threads({"affinity": 8, "effective_cores": 8.0}) == 8      # never more threads than the affinity mask allows
    assert hostinfo.baseline_threads({"affinity": 64, "effective_cores": 0.5}) == 1
