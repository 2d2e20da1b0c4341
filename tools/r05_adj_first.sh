#!/bin/bash
# round 5: first GPU pass over the fused adjoint: tests, the three legs, one rocprof summary
cd $GRAFT_REPO_ROOT
o=gpurun_out
python -m pytest tests/test_operators_gpu.py tests/test_solvers_gpu.py -m gpu -x -q -k "ptycho or full_size or headline_shapes" 2>&1 | tail -15 > $o/r05_adj_tests.txt
for w in adj256x1 adj128x1 adj256x8 fwd256x1; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $o/r05_bench_$w.json
  python - $o/r05_bench_$w.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d["config"]["workload"], "%.0f patt/s %.3f ms frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"] or 0))
PY
done > $o/r05_adj_legs.txt 2>&1
bash tools/profile_stats.sh r05 adj256x1 > /dev/null 2>&1
bash tools/profile_stats.sh r05 adj256x8 > /dev/null 2>&1
bash tools/profile_stats.sh r05 adj128x1 > /dev/null 2>&1
cat $o/r05_adj_tests.txt $o/r05_adj_legs.txt
head -8 $o/profiles_r05/r05_rocprof_stats_adj256x1.csv | cut -c1-150
head -8 $o/profiles_r05/r05_rocprof_stats_adj256x8.csv | cut -c1-150
head -8 $o/profiles_r05/r05_rocprof_stats_adj128x1.csv | cut -c1-150
