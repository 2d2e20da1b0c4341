# rocprofv3 kernel-trace summary of the default bench line (run on the GPU box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/prof_c3
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 > gpurun_out/prof_c3.log 2>&1
tail -2 gpurun_out/prof_c3.log | cut -c1-300
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -60 $f | cut -c1-170
