#!/bin/bash
# Evidence for profiles/: run on the MI355X box from the repo root.
#   bash tools/profile_round.sh r02
# Writes gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards
# (tools/profile_round.sh does the copy into gpurun_out/profiles_<tag>/).
tag=${1:-r02}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o/profiles_$tag
# 1. the bench lines
python3 bench.py > $o/profiles_$tag/${tag}_bench_c3.json 2> $o/${tag}_bench_c3.err
python3 bench.py --workload c2 --no-cpu-baseline > $o/profiles_$tag/${tag}_bench_c2.json 2>/dev/null
python3 bench.py --workload c5 --no-cpu-baseline --steps 3 > $o/profiles_$tag/${tag}_bench_c5.json 2>/dev/null
python3 bench.py --workload c1 --steps 20 --warmup 10 > $o/profiles_$tag/${tag}_bench_c1.json 2>/dev/null
for w in fwd256x1 fwd256x8 fwd128x1; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 40 --warmup 8 > $o/profiles_$tag/${tag}_bench_$w.json 2>/dev/null
done
# 2. kernel-trace summary of the default command
rm -rf $o/${tag}_prof_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_c3 -- python3 bench.py --no-cpu-baseline --no-secondary > $o/${tag}_prof_c3.log 2>&1
f=$(find $o/${tag}_prof_c3 -name "*kernel_stats.csv" | head -1)
python3 - "$f" $o/profiles_$tag/${tag}_rocprof_stats_c3.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    for r in rows[:40]:
        r[0] = r[0][:110]
        w.writerow(r[:7])
PY
# 3. HBM traffic counters, one pass per counter (no other tracing domains)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $o/${tag}_pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/${tag}_pmc_$c -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 > /dev/null 2>&1
done
python3 tools/pmc_traffic.py $o/${tag}_pmc_FETCH_SIZE $o/${tag}_pmc_WRITE_SIZE c3 1000 $o/profiles_$tag/${tag}_pmc_traffic_c3.json
# 4. the other workloads' traffic (so that every bench line's roofline has counters behind it)
bash tools/pmc_workload.sh c2 1000 $tag > /dev/null 2>&1
bash tools/pmc_workload.sh c1 256 $tag > /dev/null 2>&1
bash tools/pmc_workload.sh c5 400 $tag > /dev/null 2>&1
# 5. dynamic instruction mix of the c3 kernels (two SQ_INSTS_* passes)
bash tools/inst_mix_pmc.sh $tag > /dev/null 2>&1
cp $o/${tag}_inst_mix.md $o/profiles_$tag/${tag}_inst_mix_dynamic.md
# 6. SQ stall counters of the 128^2 forward kernel (north_star's 60 % target)
bash tools/pmc_cmd.sh fwd128_lds_kernel python3 bench.py --workload fwd128x1 --no-cpu-baseline --steps 5 --warmup 1 > $o/profiles_$tag/${tag}_pmc_sq_fwd128.txt 2>&1
# 7. the re-benched default line last (the box is warm)
python3 bench.py > $o/profiles_$tag/${tag}_bench_c3.json 2> $o/${tag}_bench_c3.err
cat $o/profiles_$tag/${tag}_bench_c3.json
