#!/bin/bash
# scripts/pmc_wgrad.sh <tag>: SQ counters of the grouped weight-gradient launch pair (scripts/probe_wgrad.py), one counter set per pass
TAG=${1:-r03}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_wgrad_pmc.txt
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_WAVES"; do
  rm -rf $OUT/pmc_w
  timeout -k 10 300 rocprofv3 --pmc $SET -d $OUT/pmc_w -o p --output-format csv -- python3 $ROOT/scripts/probe_wgrad.py > /dev/null 2>&1
  f=$(find $OUT/pmc_w -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $OUT/${TAG}_wgrad_pmc.txt <<'PY'
import csv, sys, statistics, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "wgrad" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-30:], r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "median", statistics.median(v), "n", len(v))
PY
done
rm -rf $OUT/pmc_w
cat $OUT/${TAG}_wgrad_pmc.txt
