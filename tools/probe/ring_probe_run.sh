#!/bin/bash
# Round-4 ring-probe sweep; every run bounded by its own timeout.  Output: gpurun_out/r04_ring_probe.txt
cd "$(dirname "$0")"
out=../../gpurun_out/r04_ring_probe.txt
: > $out
[ -x ./ring_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 ring_probe.hip -o ring_probe >> $out 2>&1
run() { echo "### ring_probe $*" >> $out; timeout 90 ./ring_probe "$@" >> $out 2>&1; echo "rc $?" >> $out; }
run --mode=census
run --check=1 --reps=1
run --check=1 --reps=1 --store=sc1 --load=sc1
run --check=0
run --check=0 --store=sc1 --load=sc1
run --check=0 --store=sc1 --load=acq
run --check=0 --store=nt --load=acq
run --check=0 --k=0,0,0
run --check=0 --k=0,0,0 --store=sc1 --load=sc1
run --check=0 --r1=8 --r2=8 --mode=ring
run --check=0 --r1=24 --r2=24 --mode=ring
run --check=0 --cus=72,104,80 --mode=ring
run --check=0 --cus=56,120,80 --mode=ring
run --check=0 --cus=64,128,64 --mode=ring
run --check=0 --j=2 --mode=ring
run --check=0 --j=4 --mode=ring
cat $out
