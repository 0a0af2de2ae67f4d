#!/bin/bash
# LDS bank-conflict share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) of every kernel a set of commands launches.
# usage: bash scripts/gpu_lds_conflicts.sh   (on the GPU box; output under gpurun_out/ldsc/)
export TMPDIR=/tmp
OUT=gpurun_out/ldsc; mkdir -p $OUT
i=0
while IFS= read -r cmd; do
  [ -z "$cmd" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p$i -o pmc -- $cmd > $OUT/p$i.log 2>&1 || { echo "pass $i failed: $cmd"; tail -3 $OUT/p$i.log; }
done <<'CMDS'
python3 bench.py --no-cpu-baseline --stream-log2 0 --steps 20 --warmup 5
python3 bench.py --config 1 --no-cpu-baseline --steps 20 --warmup 5
python3 bench.py --config 3 --no-cpu-baseline --steps 10 --warmup 2
python3 bench.py --config 5 --no-cpu-baseline --steps 10 --warmup 2
python3 scripts/bench_fft.py 6 8 10 11 12 13 14 15 17 20 22
python3 scripts/bench_fir.py 511 24 20
python3 scripts/bench_fir.py 32 22 20
python3 scripts/bench_chain_any.py
python3 scripts/bench_pointwise.py
CMDS
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/ldsc/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-70:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in acc.items():
    a = sum(d["SQ_LDS_IDX_ACTIVE"]) / max(1, len(d["SQ_LDS_IDX_ACTIVE"])); c = sum(d["SQ_LDS_BANK_CONFLICT"]) / max(1, len(d["SQ_LDS_BANK_CONFLICT"]))
    if a > 0: rows.append((c / a, k, a, c))
for sh, k, a, c in sorted(rows, reverse=True):
    print("%5.2f  %-70s active %.3g conflict %.3g" % (sh, k, a, c))
PY
