#!/bin/bash
# rocprofv3 kernel-trace summary of one bench workload (kernel trace only: no counters, no other domains).
#   bash tools/profile_stats.sh r04 c5 [extra bench flags]
# Writes gpurun_out/profiles_<tag>/<tag>_rocprof_stats_<workload>.csv (top 40 kernels, names cut to 110 chars).
tag=${1:-r04}; w=${2:-c5}; shift 2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o/profiles_$tag
rm -rf $o/${tag}_prof_$w
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_$w -- python3 bench.py --workload $w --no-cpu-baseline --no-secondary "$@" > $o/${tag}_prof_$w.log 2>&1
f=$(find $o/${tag}_prof_$w -name "*kernel_stats.csv" | head -1)
python3 - "$f" $o/profiles_$tag/${tag}_rocprof_stats_$w.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    for r in rows[:40]:
        r[0] = r[0][:110]
        w.writerow(r[:7])
PY
head -12 $o/profiles_$tag/${tag}_rocprof_stats_$w.csv | cut -c1-160
