run() { # env-assignment workload
  v=$(env $1 python bench.py --workload $2 --no-cpu-baseline --steps 3 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']))")
  echo "$1 $2 $v"
}
for w in c3rpie2 c5rpie2 c128rpie2; do for c in 128 256 512 1024 2048; do run TIKE_PRECOND_CHUNK=$c $w; done; done
for w in c384 c3m12 c3pad; do for c in 250 500 1000; do run TIKE_CHUNK_POSITIONS=$c $w; done; done
for w in c384 c3m12 c5; do run TIKE_STATS_GATHER=1 $w; run TIKE_EIGEN_SUMS_GATHER=0 $w; run X=1 $w; done
