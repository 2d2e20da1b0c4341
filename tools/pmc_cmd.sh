#!/bin/bash
# SQ stall counters of an arbitrary command (run on the GPU box):
#   bash tools/pmc_cmd.sh <kernel-substring> python3 bench.py --workload fwd128x1 ...
# The command after the kernel substring must be the interpreter or binary ITSELF
# on ONE GPU (python3 script.py ..., ./probe): with --pmc the profiler's preload
# initialises the GPU before the program starts, so any hop that re-executes
# (env, bash -c, taskset, numactl, a "#!/usr/bin/env" script, bench.py --gpus N
# spawning torchrun) is an exec of a GPU-initialised process -- fatal on this pool.
prog=$(basename -- "$2")
case "$prog" in
  env|bash|sh|dash|taskset|numactl|torchrun|nohup|timeout|stdbuf) echo "pmc_cmd.sh: '$2' re-executes; give the program itself" >&2; exit 2;;
esac
case "$prog" in
  python|python3|python3.*) ;;
  *) # anything else must be a native executable: a script's "#!" line is an exec too
     if [ "$(head -c 4 -- "$(command -v -- "$2" || echo "$2")" 2>/dev/null | tail -c 3)" != "ELF" ]; then
       echo "pmc_cmd.sh: '$2' is neither python3 nor an ELF binary (a script would re-exec its interpreter)" >&2; exit 2
     fi;;
esac
prev=""
for a in "$@"; do
  case "$a" in
    torch.distributed.run|torch.distributed.launch) echo "pmc_cmd.sh: a launcher re-executes the ranks" >&2; exit 2;;
    --gpus=*) [ "${a#--gpus=}" -gt 1 ] 2>/dev/null && { echo "pmc_cmd.sh: --gpus > 1 spawns ranks" >&2; exit 2; };;
  esac
  if [ "$prev" = "--gpus" ] && [ "$a" -gt 1 ] 2>/dev/null; then echo "pmc_cmd.sh: --gpus > 1 spawns ranks" >&2; exit 2; fi
  prev=$a
done
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
pat=$1; shift
o=gpurun_out/pmc_cmd
rm -rf $o
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $o -- "$@" > $o.log 2>&1
rm -rf ${o}2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d ${o}2 -- "$@" > ${o}2.log 2>&1
for d in $o ${o}2; do
f=$(find $d -name "*counter_collection.csv" | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k, v in agg.items():
    print(f"{k:28s} {v / n[k]:16.0f} per launch ({n[k]} launches)")
PY
done
