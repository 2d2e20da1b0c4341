#!/bin/bash
# SQ stall counters of the c3 minibatch kernels (run on the GPU box):
#   bash tools/pmc_kernel.sh [script args..., default tools/kbench_c3.py]  -> gpurun_out/pmc_sq/*.csv summary on stdout
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/pmc_sq
rm -rf $o
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $o -- python3 ${@:-tools/kbench_c3.py} > $o.log 2>&1
f=$(find $o -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-48:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
         "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"]
print(f"{'kernel':50s}" + "".join(f"{n[3:]:>16s}" for n in names))
for k, v in agg.items():
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k:50s}" + "".join(f"{v.get(n, 0) / wc:16.3f}" for n in names))
PY
