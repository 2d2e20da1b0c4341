#!/bin/bash
# Round-4 probe 3: is the 32 KiB row stride of the two hand-offs (rows {16 r + k1}) a DRAM-channel problem?
# Sequential launches only; row pitch / tile pitch padded.
cd "$(dirname "$0")"
out=../../gpurun_out/r04_ring_probe3.txt
: > $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 ring_probe.hip -o ring_probe >> $out 2>&1
run() { echo "### ring_probe $*" >> $out; timeout 90 ./ring_probe "$@" 2>&1 | grep -v census >> $out; echo "rc $?" >> $out; }
for k in "--k=0,0,0" "--k=33,15,12"; do
run --check=0 $k --mode=seq --reps=4
run --check=0 $k --mode=seq --reps=4 --pitch=264
run --check=0 $k --mode=seq --reps=4 --pitch=272
run --check=0 $k --mode=seq --reps=4 --pitch=288
run --check=0 $k --mode=seq --reps=4 --pitch=320
run --check=0 $k --mode=seq --reps=4 --tilepad=512
run --check=0 $k --mode=seq --reps=4 --tilepad=2048
run --check=0 $k --mode=seq --reps=4 --pitch=264 --tilepad=1024
done
run --check=1 --mode=seq --reps=1 --pitch=264 --tilepad=1024
cat $out
