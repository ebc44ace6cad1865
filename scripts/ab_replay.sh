#!/bin/bash
# scripts/ab_replay.sh "<label> ENV=VAL ..." ...: train-mode bench under each environment, prints ms/step
for spec in "$@"; do
  set -- $spec; label=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    python bench.py --mode train --no-other-modes --no-cpu-baseline > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab_$label.json").read().strip().splitlines()[-1]); print("$label", d["ms_per_step"])
except Exception as e:
    print("$label", "FAILED", e)
PY
  )
done
