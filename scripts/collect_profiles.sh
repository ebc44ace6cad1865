#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun from the repo root):
#   scripts/collect_profiles.sh <tag>      e.g. r02
# Writes gpurun_out/<tag>_*: kernel-trace summaries of the env-roofline replay, the rollout and the train step, the PMC
# (FETCH_SIZE / WRITE_SIZE, separate passes) summary of the env-step kernels, and the default bench line.
# Copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o p -- python3 $ROOT/bench.py "$@" > /dev/null 2> $OUT/${TAG}_$name.err
  local f=$(find $OUT/prof_$name -name "p_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_bench_${name}_kernel_stats.csv
  if [ "$name" = "train" ]; then
    local t=$(find $OUT/prof_$name -name "p_kernel_trace.csv" | head -1)
    [ -n "$t" ] && python3 $ROOT/scripts/timeline_update.py $t > $OUT/${TAG}_update_timeline.txt 2>&1
  fi
  rm -rf $OUT/prof_$name
  echo "[collect] $name done"
}
# (0) the default bench line (train mode, other modes, CPU baseline) — un-profiled, and it leaves TunableOp's result file
# behind, so the profiled runs below replay the chosen GEMM solutions instead of tracing the tuner's candidates
( cd $ROOT && timeout -k 10 500 python3 bench.py 2> $OUT/${TAG}_bench_line.err | tail -1 > $OUT/${TAG}_bench_line.json )
echo "[collect] bench line done"
# (1) the env roofline replay alone: 1 eager step + the 200-launch graph of macjd_env_step_timed (warm + timed replay)
stats env_roofline --mode env --steps 1 --warmup 0 --no-cpu-baseline --no-other-modes
stats rollout --mode rollout --steps 300 --warmup 100 --no-cpu-baseline --no-other-modes
stats train --mode train --steps 300 --warmup 100 --no-cpu-baseline --no-other-modes
stats env_per_env --mode env --per-env-scenarios --steps 1 --warmup 0 --no-cpu-baseline --no-other-modes
# (2) PMC passes, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass), shared and per-env tables
D=$OUT/pmc
rm -rf $D
for E in 4096 4194304; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $C -d $D/E${E}_$C -o p --output-format csv -- python3 $ROOT/bench.py --mode env --batch-envs $E --steps 30 --warmup 5 --no-cpu-baseline --no-other-modes > /dev/null 2>&1
    echo "[collect] pmc E=$E $C"
  done
done
for C in FETCH_SIZE WRITE_SIZE; do   # the rollout's many-step env launch
  timeout -k 10 300 rocprofv3 --pmc $C -d $D/MANY_$C -o p --output-format csv -- python3 $ROOT/bench.py --mode rollout --steps 300 --warmup 100 --no-cpu-baseline --no-other-modes > /dev/null 2>&1
  echo "[collect] pmc many-step $C"
done
python3 $ROOT/scripts/pmc_env_summary.py $D $TAG > $OUT/${TAG}_env_step_pmc.json
rm -rf $D
# per-env tables: ONE pass per counter at the default E = 4096; the 2^22-env launches come from bench.py's large_batch
# section (it tiles the 4096 compiled scenarios; compiling 4 M scenarios on the host would take minutes)
D=$OUT/pmc_pe
rm -rf $D
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C -d $D/ALL_$C -o p --output-format csv -- python3 $ROOT/bench.py --mode env --per-env-scenarios --steps 30 --warmup 5 --no-cpu-baseline --no-other-modes > /dev/null 2>&1
  echo "[collect] pmc per-env $C"
done
python3 $ROOT/scripts/pmc_env_summary.py $D $TAG per-env > $OUT/${TAG}_env_step_per_env_pmc.json
rm -rf $D
