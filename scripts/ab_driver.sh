#!/bin/bash
# scripts/ab_driver.sh "<label> ENV=VAL ..." ...: the driver's own invocation (--steps 20 --warmup 5) under each environment
for spec in "$@"; do
  set -- $spec; label=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/abd_$label.json 2> gpurun_out/abd_$label.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/abd_$label.json").read().strip().splitlines()[-1]); print("$label", d["ms_per_step"], d.get("modes_ms_per_step"))
except Exception as e:
    print("$label", "FAILED", e)
PY
  )
done
