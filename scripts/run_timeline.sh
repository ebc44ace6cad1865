set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; TAG=${1:-r02c}
if [ "$2" != "notests" ]; then python -m pytest tests -x -q -m gpu > $OUT/${TAG}_gpu_tests.log 2>&1; tail -1 $OUT/${TAG}_gpu_tests.log; fi
cd /tmp && export TMPDIR=/tmp
( cd $ROOT && python3 bench.py --no-cpu-baseline --no-other-modes 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json )
rm -rf $OUT/prof_tl
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tl -o p -- python3 $ROOT/bench.py --mode train --steps 300 --warmup 100 --no-cpu-baseline --no-other-modes > /dev/null 2> $OUT/${TAG}_tl.err
t=$(find $OUT/prof_tl -name "p_kernel_trace.csv" | head -1)
python3 $ROOT/scripts/timeline_update.py $t > $OUT/${TAG}_timeline.txt 2>&1
rm -rf $OUT/prof_tl
python3 -c "import json;d=json.load(open('$OUT/${TAG}_bench.json'));print(d['ms_per_step'], d['value'])"
