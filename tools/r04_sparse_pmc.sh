#!/bin/bash
# tools/r04_sparse_pmc.sh <tag> -- where one wave's time goes in a line-search round of TWO problems (one wave): cycle and instruction
# counters of wave_merit_dpp_kernel under rocprofv3 --kernel-trace --pmc, counters in their own passes
TAG=${1:-r04q}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
OUT=gpurun_out/prof_${TAG}_sparse; rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/${TAG}_sparse_pmc.txt; : > $SUM
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_SALU SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  echo "# ---- rocprofv3 --kernel-trace --pmc $ctrs -- python tools/merit_sparse.py 256 2" >> $SUM
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/p$i -o t -- python tools/merit_sparse.py 256 2 > $OUT/p$i.log 2>&1
  python tools/rocpd_summary.py $(find $OUT/p$i -name "*.db") | grep "counter\|wave_merit_dpp\|wave_merit_kernel" | cut -c1-200 >> $SUM
done
find $OUT -name "*.db" -delete
cat $SUM
