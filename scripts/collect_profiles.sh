#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run from the repo root through gpurun):
#   bash scripts/collect_profiles.sh r02
# kernel-trace statistics of the default bench workload, SQ counters and HBM traffic (separate --pmc passes) of
# scripts/one_step.py at the same batch; summaries land in gpurun_out/<tag>_* (copy the ones to keep into profiles/).
set -e
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/${tag}_kt --output-format csv -- python3 $root/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-extras > $out/${tag}_kt.log 2>&1
echo "kernel trace done" >> $out/${tag}_progress.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d $out/${tag}_pmc_a --output-format csv -- python3 $root/scripts/one_step.py 65536 3 > $out/${tag}_pmc_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT -d $out/${tag}_pmc_b --output-format csv -- python3 $root/scripts/one_step.py 65536 3 > $out/${tag}_pmc_b.log 2>&1
echo "sq counters done" >> $out/${tag}_progress.log
rocprofv3 --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch --output-format csv -- python3 $root/scripts/one_step.py 65536 3 > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/${tag}_pmc_write --output-format csv -- python3 $root/scripts/one_step.py 65536 3 > $out/${tag}_pmc_write.log 2>&1
echo "traffic done" >> $out/${tag}_progress.log
cd $root
python scripts/pmc_report.py $out/${tag}_pmc_a $out/${tag}_pmc_b > $out/${tag}_pmc_sq_counters.txt
python scripts/pmc_report.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write > $out/${tag}_pmc_traffic.txt
python scripts/pmc_traffic.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write 65536 wgs $out/${tag}_pmc_traffic.json $out/${tag}_pmc_a > /dev/null
cp $(ls $out/${tag}_kt/*/*kernel_stats.csv | head -1) $out/${tag}_train_kernel_stats.csv
echo "all done" >> $out/${tag}_progress.log
