"""Time-ordered kernel dispatches of a rocprofv3 --kernel-trace directory (all kernels): start offset, name, grid, duration, gap to the previous
dispatch's end.  usage: dispatch_timeline.py <trace dir> [first_ms] [last_ms]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[-48:], r.get("Grid_Size", "")))
rows.sort()
if not rows:
    sys.exit("no dispatches in " + sys.argv[1])
t0 = rows[0][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e30
prev_end = None
for s, e, n, g in rows:
    ms = (s - t0) / 1e6
    if lo <= ms <= hi:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print("%10.3f ms  %-48s grid=%-9s %9.4f ms   gap %8.1f us" % (ms, n, g, (e - s) / 1e6, gap))
    prev_end = max(prev_end or e, e)
