#!/bin/bash
# scripts/prof_train_mode.sh <tag> [ENV=VAL ...]: kernel-trace summary + one-update timeline of the train mode under the given environment
set -o pipefail
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o p -- python3 $ROOT/bench.py --mode train --steps 300 --warmup 100 --no-cpu-baseline --no-other-modes > $OUT/${TAG}_line.json 2> $OUT/${TAG}.err
f=$(find $OUT/prof_$TAG -name "p_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv
t=$(find $OUT/prof_$TAG -name "p_kernel_trace.csv" | head -1); [ -n "$t" ] && python3 $ROOT/scripts/timeline_update.py $t > $OUT/${TAG}_update_timeline.txt 2>&1
rm -rf $OUT/prof_$TAG
echo "[prof] $TAG done"
