#!/bin/bash
# Round 4: does the shipped library give the sub-batch gain of ab_fwdsub2.sh?  TIKE_FWD_SUB_MIB sweep
# through bench.py's own forward workloads (the library reads the variable once per process).
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_fwdsub3.txt
: > $out
run() { v=$(TIKE_FWD_SUB_MIB=$2 python3 bench.py --workload $1 --no-cpu-baseline --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f patt/s  %.3f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'] or 0))"); echo "$1 sub_mib=$2 : $v" | tee -a $out; }
for rep in 1 2; do
for mib in 0 128 192 256 320 384; do run fwd256x1 $mib; done
for mib in 0 256; do run fwd256x8 $mib; done
done
