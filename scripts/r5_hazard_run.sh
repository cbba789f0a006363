#!/bin/bash
# round 5, experiment 1: the MFMA chain hazard in isolation, then round 4's failing case on the library with the 16-deep tail
# (bug) and with the 32-deep zero-padded tail (fix)
set -o pipefail
mkdir -p gpurun_out/r5
hipcc -O3 --offload-arch=gfx950 scripts/microbench/mfma_chain_hazard.hip -o /tmp/mfma_chain_hazard 2>/dev/null && /tmp/mfma_chain_hazard > gpurun_out/r5/hazard.txt 2>&1
cat gpurun_out/r5/hazard.txt
for lib in A_bug 4_3_5_2_61_40_70_12_24; do
  for i in 1 2 3 4 5 6; do
    echo "== $lib run $i" >> gpurun_out/r5/split_ab.txt
    PMT_LIB=$PWD/permutect_amd/instances/libpermutect_amd_$lib.so NREF='[10]' NALT='[300]' timeout -k 10 300 python scripts/wide_debug2.py A 2>&1 | grep -E "shape id|grad rel|total rel|fault" >> gpurun_out/r5/split_ab.txt
  done
done
cat gpurun_out/r5/split_ab.txt
