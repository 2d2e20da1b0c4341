#!/bin/bash
# round 5: A/B of grouped-scatter builds on one box: c3 epoch kernel table + adjoint legs per library
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05_ab_scatter.txt; : > $o
for l in "$@"; do
  for w in adj256x1 adj128x1; do
    TIKE_AMD_LIB=$PWD/tools/probe/_lib/lib_$l.so python3 bench.py --workload $w --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %-9s %.0f patt/s %.3f ms frac %.3f' % ('$l', '$w', d['value'], d['ms_per_step'], d['roofline']['frac']))" | tee -a $o
  done
done
WL=c3 STEPS=5 bash tools/ab_libs.sh 1 "$@" | tee -a $o
WL=c2 STEPS=3 bash tools/ab_libs.sh 1 "$@" | tee -a $o
