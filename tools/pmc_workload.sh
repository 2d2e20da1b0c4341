#!/bin/bash
# HBM traffic counters of one bench workload: bash tools/pmc_workload.sh <workload> <positions per launch> <tag>
w=$1; n=$2; tag=${3:-r02}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o/profiles_$tag
# (operator workloads: the positions of the leg the file is quoted for)
pos=""; case $w in fwd*|adj*) pos="--positions $n";; esac
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $o/${tag}_pmc_${w}_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/${tag}_pmc_${w}_$c -- python3 bench.py --workload $w $pos --no-cpu-baseline --no-secondary --steps 1 --warmup 0 > /dev/null 2>&1
done
python3 tools/pmc_traffic.py $o/${tag}_pmc_${w}_FETCH_SIZE $o/${tag}_pmc_${w}_WRITE_SIZE $w $n $o/profiles_$tag/${tag}_pmc_traffic_$w.json
