#!/bin/bash
# PMC passes over scripts/bench_fir.py (one algo), each counter group in its own run.
# usage: bash scripts/gpu_pmc.sh <tag> <algo> [n_taps]
TAG=${1:-pmc}; ALGO=${2:-os1024}; TAPS=${3:-255}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  ALGOS=$ALGO timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 scripts/bench_fir.py $TAPS 24 20 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "fir_" in r["Kernel_Name"] and "hist" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print("%-28s per-launch mean %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
