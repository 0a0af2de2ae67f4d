"""Per-kernel totals from a rocprofv3 --kernel-trace CSV: calls, average and summed duration, by kernel name (template
arguments kept).  usage: python scripts/kstats.py <dir-or-csv> [substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(list)
for r in csv.DictReader(open(p)):
    name = r["Kernel_Name"].split("(")[0].replace("void comms::", "")
    if sub in name:
        acc[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-70s calls %5d  avg %9.1f us  median %9.1f  sum %10.1f us" % (name[:70], len(v), sum(v) / len(v), v2[len(v2) // 2], sum(v)))
