#!/bin/bash
# EXPERIMENT: the one-tile-per-wave backward (csrc/ab/lib_rt1.so) -- parity on the default library's fixtures, then an A/B of the train step
mkdir -p gpurun_out/r5 permutect_amd/csrc/ab
export PMT_JIT=0
cp -f permutect_amd/libpermutect_amd_rt1.so permutect_amd/csrc/ab/lib_rt1.so 2>/dev/null
PMT_LIB=$PWD/permutect_amd/csrc/ab/lib_rt1.so timeout -k 10 500 python -m pytest tests/test_train_gpu.py -q -x -k "p0_b16 or p0_zero_ref or p0_saturated or p0_deep or random_mixed or beyond_one or layered or zero_adversarial" > gpurun_out/r5/rt1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5/rt1_tests.log
tail -5 gpurun_out/r5/rt1_tests.log
timeout -k 10 500 bash scripts/ab_bwd.sh base rt1 > gpurun_out/r5/rt1_ab.log 2>&1
cat gpurun_out/r5/rt1_ab.log
