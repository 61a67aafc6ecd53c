#!/bin/bash
# Developer helper: the same quick_bench lines with two builds of the library on ONE box (boxes differ by 8 %):
#   ab_tmp/old.so, ab_tmp/new.so (untracked; built by hand from two states of the tree), alternating old/new/old/new.
set -e
mkdir -p gpurun_out
for round in 1 2; do
  for v in old new; do
    cp ab_tmp/$v.so cova_amd/libcovahip.so
    for lanes in 1 2; do
      echo "== $v lanes=$lanes round=$round"
      QB_INPUT=frames QB_LANES=$lanes timeout -k 10 120 python tools/quick_bench.py 256 300
    done
  done
done
