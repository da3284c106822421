"""Start / end of every Gibbs kernel dispatch from a rocprofv3 --kernel-trace directory.  usage: timeline.py <trace dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gibbs" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), ("simple" if "gibbs_simple" in r["Kernel_Name"] else "hot" if "gibbs_hot" in r["Kernel_Name"] else "general"), r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("LDS_Block_Size", "")))
rows.sort()
if not rows:
    print("no gibbs dispatches found in", sys.argv[1])
    sys.exit(0)
t0 = rows[0][0]
for s, e, k, g, w, l in rows:
    if (e - s) / 1e6 > 1.0:
        print("%9.1f ms -> %9.1f ms  (%8.1f ms)  %s grid %s wg %s lds %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, k, g, w, l))
