"""Print a rocprofv3 kernel_stats.csv with short kernel names.  usage: kstats.py <dir>"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "rocprim" in n:
            i = n.find("onesweep")
            short = "rocprim " + ("onesweep " if i >= 0 else "") + n[-70:]
        else:
            short = n.replace("(anonymous namespace)::", "").split("(")[0][-70:]
        print("%-80s calls %5s avg %10.3f ms total %9.1f ms" % (short[:80], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
