#!/bin/bash
# PMC passes over one bench step (contexts=1); each counter group in its own run.
# usage: gpu_pmc.sh <outdir-tag> [env assignments]
tag=$1; shift
mkdir -p gpurun_out/pmc_$tag; export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
BENCH="env $* python $R/bench.py --steps 1 --warmup 0 --contexts 1 --streams 256 --no-cpu-baseline --no-check"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_IFETCH SQ_INSTS_VSKIPPED" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --stats -d $R/gpurun_out/pmc_$tag/g$i -o g$i --output-format csv -- $BENCH > $R/gpurun_out/pmc_$tag/g$i.log 2>&1
  echo "group $i rc=$?"
done
