#!/bin/bash
# development aid: A/B of per-translation-unit flags, several units per variant:  bash scripts/ab3.sh "tu1:flags;tu2:flags" ...
# (each variant: touch + rebuild the named units with FLAGS_<tu>, then scripts/kernel_times.py)
set -e
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v" | tee -a gpurun_out/ab.log
  args=""
  IFS=';' read -ra parts <<< "$v"
  for part in "${parts[@]}"; do
    tu="${part%%:*}"; fl="${part#*:}"
    touch permutect_amd/csrc/$tu.hip
    args="$args FLAGS_$tu=\"$fl\""
  done
  eval make -C permutect_amd/csrc -j12 $args > gpurun_out/ab_build.log 2>&1
  python scripts/kernel_times.py 65536 10 2>&1 | grep "KT" | tee -a gpurun_out/ab.log
done
