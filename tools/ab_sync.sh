cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for x in 0 1; do
v=$(TIKE_X_NO_COST_SYNC=$x python3 bench.py --no-cpu-baseline --no-secondary --steps 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.3f %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_time_share_of_wall']))")
echo "nosync=$x $v"; done; done
