#!/bin/bash
# usage: run_variants.sh name...  -> error table for each build_variants/<name>.so
mkdir -p gpurun_out
for v in "$@"; do
  MOD16_LIB=build_variants/$v.so timeout -k 10 200 python bench.py --dtype float32 --math mixed --no-cpu-baseline --no-plain --no-configs --steps 5 > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err || exit 1
done
