"""What the host of this box gives a process, and how the oracle's threaded Gibbs leg (bench.py's cpu_baseline) scales on it.
Prints one JSON object: cpu_count, affinity, cgroup quota, a spin calibration (tools/cpu_spin.c: N threads of register-only work; effective
cores = N x t(1) / t(N)) and cluster-sweeps/s of ONE fixed sample of the bench mixture at 1 / 8 / 32 / 64 / 128 / all threads.
usage: python tools/cpu_scaling.py [S] [groups] [out.json]     (TEST / BENCH infrastructure: uses oracle/)"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def host_facts():
    f = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            f[p] = open(p).read().strip()
        except OSError:
            pass
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                f["model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        f["loadavg"] = open("/proc/loadavg").read().split()[:3]
    except OSError:
        pass
    q = f.get("/sys/fs/cgroup/cpu.max", "max").split()
    f["cgroup_cores"] = None if q[0] == "max" else float(q[0]) / float(q[1])
    return f


def spin_curve(counts, iters=300_000_000):
    exe = os.path.join(tempfile.gettempdir(), "bt_cpu_spin")
    subprocess.check_call(["gcc", "-O2", "-pthread", os.path.join(ROOT, "tools", "cpu_spin.c"), "-o", exe])
    out = {}
    t1 = None
    for n in counts:
        t = float(subprocess.check_output([exe, str(n), str(iters)]).decode())
        t1 = t if t1 is None else t1
        out[str(n)] = {"s": t, "effective_cores": n * t1 / t}
    return out


def thread_counts(top):
    c = [n for n in (1, 8, 32, 64, 128) if n < top]
    return c + [top]


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    import _oracle
    from bayestyper_amd import synth
    facts = host_facts()
    top = facts["affinity"]
    facts["spin"] = spin_curve(thread_counts(top))
    facts["effective_cores"] = max(v["effective_cores"] for v in facts["spin"].values())
    orc = _oracle.load_oracle()
    lg, ln = _oracle.build_luts(orc, S)
    flat = synth.make_mixture(G, S, seed=999)
    curve = {}
    for n in reversed(thread_counts(top)):   # all threads first: the one-thread leg is the long one and can be cut short
        # the one-thread leg runs every k-th group of the sample (same class mix) so that it takes seconds, not minutes
        from bayestyper_amd import shard
        import numpy as np
        stride = max(1, int(round(min(top, 64) / max(n, 1) / 2))) if n < 64 else 1
        f = shard.take_groups(flat, np.arange(0, flat["num_groups"], stride)) if stride > 1 else flat
        og = _oracle.OrcGibbs(orc, f, lg, ln, seed=42)
        t = time.perf_counter()
        og.run(n)
        dt = time.perf_counter() - t
        og.close()
        curve[str(n)] = {"groups": int(f["num_groups"]), "s": dt, "cluster_sweeps_per_s": f["num_clusters"] * 7000 / dt}
        print(n, curve[str(n)], file=sys.stderr, flush=True)
    one = curve["1"]["cluster_sweeps_per_s"]
    for v in curve.values():
        v["over_one_thread"] = v["cluster_sweeps_per_s"] / one
    facts["oracle_gibbs"] = {"S": S, "groups": G, "mixture": flat["mixture"], "curve": curve}
    text = json.dumps(facts, indent=1)
    print(text)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")


if __name__ == "__main__":
    main()
