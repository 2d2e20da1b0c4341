#!/bin/bash
# A/B of library builds on the operator legs: tools/ab_fwd.sh <reps> "<workloads>" <lib> ...
reps=$1; ws=$2; shift 2
for rep in $(seq $reps); do
  for l in "$@"; do
    for w in $ws; do
      TIKE_AMD_LIB=$PWD/tools/probe/_lib/lib_$l.so python3 bench.py --workload $w --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %-9s %.0f patt/s %.4f ms frac %.3f' % ('$l', '$w', d['value'], d['ms_per_step'], d['roofline']['frac']))"
    done
  done
done
