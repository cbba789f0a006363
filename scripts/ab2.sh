#!/bin/bash
# development aid: A/B of the backward translation unit's extra flags (FLAGS_pmt_backward), alternating builds on one box
set -e
mkdir -p gpurun_out
for v in "$@"; do
  echo "== FLAGS_pmt_backward='$v'" | tee -a gpurun_out/ab.log
  touch permutect_amd/csrc/pmt_backward.hip
  make -C permutect_amd/csrc -j12 FLAGS_pmt_backward="$v" > gpurun_out/ab_build.log 2>&1
  python scripts/kernel_times.py 65536 10 2>&1 | grep KT | tee -a gpurun_out/ab.log
done
