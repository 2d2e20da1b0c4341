#!/bin/bash
# Dynamic instruction mix of the c3 minibatch kernels (run on the GPU box, one GPU):
#   bash tools/inst_mix_pmc.sh [tag]   -> gpurun_out/<tag>_inst_mix.md
# Two --pmc passes of the same command (8 SQ slots each), --kernel-trace only.
# The program after `--` is the interpreter itself (no env / bash -c hop).
tag=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/${tag}_instmix
rm -rf ${o}_a ${o}_b
cmd="python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d ${o}_a -- $cmd > ${o}_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d ${o}_b -- $cmd > ${o}_b.log 2>&1
python3 tools/inst_mix_pmc.py ${o}_a ${o}_b > gpurun_out/${tag}_inst_mix.md
cat gpurun_out/${tag}_inst_mix.md
