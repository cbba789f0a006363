#!/bin/bash
# development aid: build the library with each EXTRA flag set given (one per argument, "" = default) and time the kernels
set -e
mkdir -p gpurun_out
for v in "$@"; do
  echo "== EXTRA='$v'" | tee -a gpurun_out/ab.log
  touch permutect_amd/csrc/*.hip
  make -C permutect_amd/csrc -j12 EXTRA="$v" > gpurun_out/ab_build.log 2>&1
  python scripts/kernel_times.py 65536 10 2>&1 | tee -a gpurun_out/ab.log
done
