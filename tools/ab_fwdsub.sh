#!/bin/bash
# Round-4 experiment: forward operator at 256^2 in sub-batches (TIKE_FWD_SUB positions) so that the
# in-place column pass finds its input in the Infinity Cache; nt vs plain hand-off stores.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_fwdsub.txt
: > $out
for wl in fwd256x1 fwd256x8; do
for lib in base plainf1; do
for sub in 0 512 256 192 128 96 64 32; do
  [ $wl = fwd256x8 ] && sub=$((sub / 8))
  v=$(TIKE_FWD_SUB=$sub TIKE_AMD_LIB=$PWD/tools/probe/_lib/lib_$lib.so python3 bench.py --workload $wl --no-cpu-baseline --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f patt/s  %.3f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'] or 0))")
  echo "$wl lib=$lib sub=$sub : $v" | tee -a $out
done; done; done
