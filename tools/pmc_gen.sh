#!/bin/bash
# SQ stall / instruction counters of the shape-general kernels on the c384 workload (GPU box):
#   bash tools/pmc_gen.sh  -> summary on stdout (two --pmc passes, --kernel-trace only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cmd="python3 bench.py --workload c384 --positions 1000 --no-cpu-baseline --no-secondary --steps 1 --warmup 0"
for pass in a b; do
  o=gpurun_out/pmc_gen_$pass
  rm -rf $o
  if [ $pass = a ]; then
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $o -- $cmd > $o.log 2>&1
  else
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $o -- $cmd > $o.log 2>&1
  fi
  f=$(find $o -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-44:]
    if "gen_" not in k and "mix_" not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({n for v in agg.values() for n in v})
print(f"{'kernel':46s}" + "".join(f"{n[3:]:>18s}" for n in names))
for k, v in agg.items():
    print(f"{k:46s}" + "".join(f"{v.get(n, 0):18.4g}" for n in names))
PY
done
