#!/bin/bash
# round 5: operator legs + short epoch workloads, one line each -> gpurun_out/r05_legs_<tag>.txt
#   bash tools/r05_legs.sh <tag> [workloads...]
cd $GRAFT_REPO_ROOT
o=gpurun_out; tag=${1:-x}; shift
ws=${@:-adj256x1 adj128x1 adj256x8 fwd256x1}
for w in $ws; do
  python bench.py --workload $w --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $o/r05_bench_${w}_$tag.json
  python - $o/r05_bench_${w}_$tag.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]
print(d["config"]["workload"], "%.0f patt/s %.3f ms/step | %s %.3f ms frac %s" % (d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"]))
PY
done > $o/r05_legs_$tag.txt 2>&1
cat $o/r05_legs_$tag.txt
