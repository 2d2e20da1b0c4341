#!/bin/bash
# Round-6 evidence for profiles/ (run on the MI355X box from the repo root):
#   bash tools/profile_round6.sh r06 [part]      part: bench | stats | pmc | all
# Writes gpurun_out/profiles_<tag>/; copy what is to be judged into profiles/.
tag=${1:-r06}; part=${2:-all}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o/profiles_$tag
if [ $part = bench ] || [ $part = all ]; then
  python3 bench.py > $o/profiles_$tag/${tag}_bench_c3.json 2> $o/${tag}_bench_c3.err
  python3 bench.py --workload c1 --steps 20 --warmup 10 > $o/profiles_$tag/${tag}_bench_c1.json 2>/dev/null
  for w in c2 c5 c3poisson c3rpie c3rpie2 c3pad c3m12 c384 c5rpie2 c128rpie2 c3rpie2poisson; do
    python3 bench.py --workload $w --no-cpu-baseline --steps 3 > $o/profiles_$tag/${tag}_bench_$w.json 2>/dev/null
  done
  for w in fwd256x1 fwd256x8 fwd128x1 adj256x1 adj256x8 adj128x1; do
    python3 bench.py --workload $w --no-cpu-baseline --steps 40 --warmup 8 > $o/profiles_$tag/${tag}_bench_$w.json 2>/dev/null
  done
fi
if [ $part = stats ] || [ $part = all ]; then
  for w in c3 c1 c2 c5 c3poisson c3rpie c3rpie2 c384 c3m12 c3pad c5rpie2 adj256x1 adj256x8 adj128x1; do
    extra=""; case $w in c2|c5|c3poisson|c3rpie2|c384|c3m12|c3pad|c5rpie2) extra="--steps 3";; esac
    bash tools/profile_stats.sh $tag $w $extra > /dev/null 2>&1
  done
fi
if [ $part = pmc ] || [ $part = all ]; then
  bash tools/pmc_workload.sh c3 1000 $tag > /dev/null 2>&1
  bash tools/pmc_workload.sh c2 1000 $tag > /dev/null 2>&1
  bash tools/pmc_workload.sh c1 256 $tag > /dev/null 2>&1
  bash tools/pmc_workload.sh c5 1000 $tag > /dev/null 2>&1
  bash tools/pmc_workload.sh adj256x1 4096 $tag > /dev/null 2>&1
  bash tools/pmc_workload.sh adj256x8 512 $tag > /dev/null 2>&1
  bash tools/pmc_workload.sh adj128x1 16384 $tag > /dev/null 2>&1
  bash tools/pmc_workload.sh fwd256x1 4096 $tag > /dev/null 2>&1
  bash tools/pmc_workload.sh fwd256x8 512 $tag > /dev/null 2>&1
  bash tools/pmc_workload.sh fwd128x1 16384 $tag > /dev/null 2>&1
  bash tools/pmc_workload.sh c3poisson 1000 $tag > /dev/null 2>&1
  bash tools/pmc_workload.sh c384 1000 $tag > /dev/null 2>&1
fi
ls $o/profiles_$tag | wc -l
