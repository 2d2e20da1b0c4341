#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_fwdsub2.txt
: > $out
run() { v=$(TIKE_FWD_SUB=$3 TIKE_AMD_LIB=$PWD/tools/probe/_lib/lib_$2.so python3 bench.py --workload $1 --no-cpu-baseline --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f patt/s  %.3f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'] or 0))"); echo "$1 lib=$2 sub=$3 : $v" | tee -a $out; }
for rep in 1 2; do
run fwd256x1 base 0
for sub in 384 512 640 768 1024 2048; do run fwd256x1 plainf1 $sub; done
run fwd256x8 base 0
for sub in 48 64 96 128 256; do run fwd256x8 plainf1 $sub; done
done
