"""Durations of one kernel's launches in launch order from a rocprofv3 --kernel-trace CSV, in groups of ten.
usage: python3 scripts/kseq.py <dir> <kernel-substring>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
d = [(b - a) / 1e3 for a, b in rows]
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
print("%d launches; mean %.1f us, median %.1f, min %.1f, max %.1f; median gap %.1f us" % (len(d), sum(d) / len(d), sorted(d)[len(d) // 2], min(d), max(d), sorted(gaps)[len(gaps) // 2] if gaps else 0))
for i in range(0, len(d), 10):
    print("%4d: %s" % (i, " ".join("%6.1f" % v for v in d[i:i + 10])))
