"""The 2^30-sample stream leg of `bench.py` launch by launch, from a rocprofv3 kernel trace: every pass of the
node-by-node chain whose FIR launch is longer than 1 ms (the headline leg's are ~50 us), with the gap in front of
it and the mixer / decimate kernels that follow.
usage: python scripts/trace_stream_leg.py <kernel_trace.csv>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
t0 = int(rows[0]["Start_Timestamp"])
n = 0
for i, r in enumerate(rows):
    if "fir_os1024_dyn" not in r["Kernel_Name"] or dur(r) < 1.0:
        continue
    gap = (int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i else float("nan")
    nxt = [(q["Kernel_Name"].split("(")[0].split("<")[0][-24:], dur(q)) for q in rows[i + 1:i + 3]]
    print("pass %2d at %9.3f ms: gap before %8.1f us | fir %.3f ms | %s" % (
        n, (int(r["Start_Timestamp"]) - t0) / 1e6, gap, dur(r), " | ".join("%s %.3f ms" % x for x in nxt)))
    n += 1
print("%d stream-leg FIR launches" % n)
