#!/bin/bash
# A/B of backward-kernel variants on ONE box: each library in turn, twice (alternating), the bench's train-only leg
mkdir -p gpurun_out/r5
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = "base" ]; then lib=""; else lib="PMT_LIB=$PWD/permutect_amd/csrc/ab/lib_$v.so"; fi
    line=$(env $lib python bench.py --mode train --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep "train:")
    echo "$v $round: $line" | sed 's/\[bench *[0-9.]*s\] //' | cut -c1-170
  done
done
