#!/bin/bash
# Same-device A/B of builds of the library on the headline step: tools/ab_bench.sh OUT LIB... 
# (each LIB twice, interleaved; prints ms_per_step and the kernel's HIP-event time). Extra bench
# arguments through AB_ARGS, e.g. AB_ARGS="--dtype float32 --math mixed".
out=$1; shift
: > "$out"
for round in 1 2; do
  for lib in "$@"; do
    MOD16_LIB=$lib timeout -k 10 200 python bench.py --steps 40 --no-configs --no-parity --no-cpu-baseline --no-plain $AB_ARGS 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'ms_per_step %.3f kernel_ms %.3f frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))" >> "$out" || exit 1
  done
done
cat "$out"
