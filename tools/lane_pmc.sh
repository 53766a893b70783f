#!/bin/bash
# dynamic instruction counts of the LANE iLQR kernels for ONE wave (batch 64): bash tools/lane_pmc.sh -> gpurun_out/lane_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lane_pmc.txt; : > $out
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" "SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs -d gpurun_out/lp_$i -o t -- python tools/lane_latency.py 64 1 1 > gpurun_out/lp_$i.log 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/lp_$i -name "*.db" | head -1) | grep -v "rocclr\|pack\|copy" | cut -c1-200 >> $out
done
