#!/bin/bash
# LDS counters only (one rocprofv3 --pmc pass) over a command; per-launch means of the kernels whose name contains <kernel-substring>.
# usage: bash scripts/gpu_pmc_lds.sh <tag> <kernel-substring> python3 script.py args...
TAG=$1; KSUB=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/p -o pmc -- "$@" > $OUT/p.log 2>&1 || { echo "pass failed"; tail -3 $OUT/p.log; }
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/p/*counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    ks = sorted({k for k, _ in acc})
    for kn in ks:
        g = lambda c: sum(acc[(kn, c)]) / max(1, len(acc[(kn, c)]))
        print("%-48s LDS cycles %.4g, bank-conflict cycles %.4g (%.0f %%), LDS instructions %.4g, wave cycles %.4g" % (kn, g("SQ_LDS_IDX_ACTIVE"), g("SQ_LDS_BANK_CONFLICT"), 100 * g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE")), g("SQ_INSTS_LDS"), g("SQ_WAVE_CYCLES")))
PY
