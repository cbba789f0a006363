#!/bin/bash
# A/B of forward-kernel variants on ONE box: each library in turn, twice (alternating), the bench's filter and train legs
# (csrc/ab/lib_<name>.so through PMT_LIB; "base" = the shipped library)
mkdir -p gpurun_out/r5
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = "base" ]; then lib=""; else lib="PMT_LIB=$PWD/permutect_amd/csrc/ab/lib_$v.so"; fi
    out=$(env $lib python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>&1 >/dev/null | grep "train:\|filter:")
    echo "$out" | sed "s/^\[bench *[0-9.]*s\] /$v $round: /" | cut -c1-200
  done
done
