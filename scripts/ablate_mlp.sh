#!/bin/bash
# Builds scripts/ablate_mlp.cpp once per ablation mask into scratch/ablate/ (git-ignored; travels with gpurun) —
# run the binaries on the GPU box:  for b in scratch/ablate/mlp_*; do $b; done
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/ablate
for m in 0 1 2 3 4 16 20 23; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Wno-pass-failed -x hip \
      -DMACJD_MLP_ABLATE=$m scripts/ablate_mlp.cpp -o scratch/ablate/mlp_$m &
done
wait
ls scratch/ablate
