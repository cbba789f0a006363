#!/bin/bash
# development aid: A/B of ONE translation unit's extra flags (FLAGS_<tu> in csrc/Makefile), alternating builds on one box:
#   bash scripts/ab2.sh pmt_backward "" "-mllvm -amdgpu-sched-strategy=max-ilp"
set -e
tu=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  echo "== FLAGS_$tu='$v'" | tee -a gpurun_out/ab.log
  touch permutect_amd/csrc/$tu.hip
  make -C permutect_amd/csrc -j12 FLAGS_$tu="$v" > gpurun_out/ab_build.log 2>&1
  python scripts/kernel_times.py 65536 10 2>&1 | grep "KT\|rows_backward\|cnn3_backward\|cnn3_forward" | tee -a gpurun_out/ab.log
done
