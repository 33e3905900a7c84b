#!/bin/bash
# PMC passes over one bench step (contexts=1); each counter group in its own run.
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
BENCH="python $R/bench.py --steps 1 --warmup 0 --contexts 1 --no-cpu-baseline --no-check"
rocprofv3 --list-avail > $R/gpurun_out/pmc/avail.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --stats -d $R/gpurun_out/pmc/g$i -o g$i --output-format csv -- $BENCH > $R/gpurun_out/pmc/g$i.log 2>&1
  echo "group $i rc=$?"
done
ls -R $R/gpurun_out/pmc | head -50
