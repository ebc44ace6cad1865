#!/bin/bash
# scripts/ab_updates.sh "ENV=VAL ..." "ENV=VAL ..." ...: scripts/time_updates.py under each environment ("-" = defaults), twice, interleaved
for round in 1 2; do
  for spec in "$@"; do
    ( [ "$spec" != "-" ] && for kv in $spec; do export "$kv"; done
      timeout -k 10 200 python scripts/time_updates.py 2000 4 2>&1 | grep "us/update" )
  done
done
