#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the ring probe's kernels (separate --pmc passes, kernel trace only):
# the sequential launches only -- counter collection serialises dispatches, so the three
# resident-together kernels cannot run under it (their bounded spins would time out); their
# loads and stores are the same instructions on the same bytes.  Output: gpurun_out/r04_ring_pmc.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04_ring_pmc.txt
: > $out
[ -x tools/probe/ring_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe/ring_probe.hip -o tools/probe/ring_probe
for mode in seq; do
for k in 33,15,12; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/ring_pmc_${mode}_${ctr}
  rm -rf $d
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- ./tools/probe/ring_probe --check=0 --k=$k --store=sc1 --load=sc1 --cus=80,112,64 --mode=$mode --reps=2 > $d.log 2>&1
  echo "### mode $mode k $k $ctr" >> $out
  grep -E "sequential|resident together" $d.log | tail -1 >> $out
  python3 tools/pmc_sum.py $d kern >> $out
done; done; done
cat $out
