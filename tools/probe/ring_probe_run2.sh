#!/bin/bash
# Round-4 ring-probe sweep 2: VALU load calibrated to the product kernels' measured VALU time
# (0.545 / 0.752 / 0.407 ms per 1000 positions at full chip), write-through stores + sc1 loads.
cd "$(dirname "$0")"
out=../../gpurun_out/r04_ring_probe2.txt
: > $out
[ -x ./ring_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 ring_probe.hip -o ring_probe >> $out 2>&1
run() { echo "### ring_probe $*" >> $out; timeout 90 ./ring_probe "$@" 2>&1 | grep -v census >> $out; echo "rc $?" >> $out; }
K=--k=33,15,12
run --check=0 $K --store=sc1 --load=sc1 --cus=80,112,64
run --check=0 $K --store=sc1 --load=sc1 --cus=88,104,64 --mode=ring
run --check=0 $K --store=sc1 --load=sc1 --cus=72,120,64 --mode=ring
run --check=0 $K --store=sc1 --load=sc1 --cus=80,104,72 --mode=ring
run --check=0 $K --store=sc1 --load=sc1 --cus=96,96,64 --mode=ring
run --check=0 $K --store=sc1 --load=sc1 --cus=80,112,64 --r1=8 --r2=8 --mode=ring
run --check=0 $K --store=sc1 --load=sc1 --cus=80,112,64 --r1=24 --r2=24 --mode=ring
run --check=0 $K --store=sc1 --load=sc1 --cus=80,112,64 --r1=32 --r2=32 --mode=ring
run --check=0 $K --store=sc1 --load=sc1 --cus=80,112,64 --j=2 --mode=ring
run --check=0 $K --store=sc1 --load=sc1 --cus=80,112,64 --j=4 --mode=ring
run --check=0 --k=16,8,6 --store=sc1 --load=sc1 --cus=80,112,64
run --check=0 $K --store=plain --load=sc1 --cus=80,112,64 --mode=ring
run --check=1 $K --store=sc1 --load=sc1 --cus=80,112,64 --mode=ring --reps=2
cat $out
