"""Per-step kernel durations and gaps of the headline chain from a rocprofv3 kernel trace (the last K steps).
usage: python scripts/trace_steps.py <kernel_trace.csv> [K]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# steps = triples fir_os1024_dyn -> mixer -> decimate, consecutive
idx = [i for i in range(len(rows) - 2) if "fir_os1024_dyn" in names[i] and "mixer_kernel" in names[i + 1] and "decimate_kernel" in names[i + 2]]
# the headline leg's steps are the last W + K triples that work on 2^24 samples (the stream leg's kernels are far longer)
short = [i for i in idx if int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) < 100000]
sel = short[-K:]
print("steps found %d (headline-size %d), showing the last %d" % (len(idx), len(short), len(sel)))
prev_end = None
tot = []
for n, i in enumerate(sel):
    s = [int(rows[i + j]["Start_Timestamp"]) for j in range(3)]
    e = [int(rows[i + j]["End_Timestamp"]) for j in range(3)]
    d = [(e[j] - s[j]) / 1e3 for j in range(3)]
    g = [(s[1] - e[0]) / 1e3, (s[2] - e[1]) / 1e3]
    lead = (s[0] - prev_end) / 1e3 if prev_end else float("nan")
    prev_end = e[2]
    tot.append(d[0] + d[1] + d[2] + g[0] + g[1] + (lead if lead == lead else 0))
    print("step %2d: gap before %6.1f | fir %5.1f gap %4.1f mixer %5.1f gap %4.1f decimate %5.1f us" % (n, lead, d[0], g[0], d[1], g[1], d[2]))
print("mean period %.1f us" % (sum(tot[1:]) / max(1, len(tot) - 1)))
