#!/bin/bash
# tools/r04_al_pmc.sh <tag> -- instruction counters of the constrained (12, 4) merit kernels: two whole C1 + input-bound solves under
# rocprofv3 --kernel-trace --pmc (counters in their own passes); per-dispatch mean / min / max (the full-batch launches are the max)
TAG=${1:-r04m}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
OUT=gpurun_out/prof_${TAG}_alpmc; rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/${TAG}_al_merit_pmc.txt; : > $SUM
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  echo "# ---- rocprofv3 --kernel-trace --pmc $ctrs -- python tools/c1_solve.py 2 4096 256 --al" >> $SUM
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/p$i -o t -- python tools/c1_solve.py 2 4096 256 --al > $OUT/p$i.log 2>&1
  python tools/rocpd_summary.py $(find $OUT/p$i -name "*.db") | grep "counter\|wave_merit_dpp\|mfma16_backward\|wave_expand_dpp" | cut -c1-200 >> $SUM
done
find $OUT -name "*.db" -delete
cat $SUM
