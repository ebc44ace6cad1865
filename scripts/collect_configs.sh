#!/bin/bash
# scripts/collect_configs.sh <tag>: bench lines + train-mode kernel-trace summaries of BASELINE.json configs 3 and 5 (per GPU)
set -o pipefail
TAG=${1:-r03}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
run() {  # name, bench args...
  local name=$1; shift
  ( cd $ROOT && timeout -k 10 500 python3 bench.py "$@" 2> $OUT/${TAG}_${name}_line.err | tail -1 > $OUT/${TAG}_${name}_line.json )
  echo "[configs] $name line done"
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/prof_$name && \
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o p -- python3 $ROOT/bench.py "$@" --mode train --steps 300 --warmup 100 --no-cpu-baseline --no-other-modes > /dev/null 2> $OUT/${TAG}_${name}_prof.err )
  local f=$(find $OUT/prof_$name -name "p_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_${name}_train_kernel_stats.csv
  local t=$(find $OUT/prof_$name -name "p_kernel_trace.csv" | head -1); [ -n "$t" ] && python3 $ROOT/scripts/timeline_update.py $t > $OUT/${TAG}_${name}_update_timeline.txt 2>&1
  rm -rf $OUT/prof_$name
  echo "[configs] $name profile done"
}
# ONLY="c5" scripts/collect_configs.sh <tag>: just the configurations whose name contains $ONLY
want() { [ -z "$ONLY" ] || [[ "$1" == *"$ONLY"* ]]; }
want c3_6j8r && run c3_6j8r --jammers 6 --radars 8
want c5_12j16r_fp32 && run c5_12j16r_fp32 --jammers 12 --radars 16 --batch-envs 2048
want c5_12j16r_bf16 && run c5_12j16r_bf16 --jammers 12 --radars 16 --batch-envs 2048 --mixer-dtype bf16
true
