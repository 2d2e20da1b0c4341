#!/bin/bash
# Same-box A/B of an environment switch: tools/ab_env.sh <reps> VAR=a VAR=b ...
# alternating runs of the default bench (STEPS, WL as in ab_libs.sh).
reps=$1; shift
for rep in $(seq $reps); do
  for kv in "$@"; do
    v=$(env $kv python3 bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-10} ${WL:+--workload $WL} 2>/dev/null |
      python3 -c "import sys, json; print('%.0f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$kv $v"
  done
done
