#!/bin/bash
# tools/r03_merit2_pmc.sh <tag>: PMC counters of the two forms of the two-trial merit evaluation on C1 (one pass per counter set,
# --kernel-trace only): HBM bytes (FETCH_SIZE, WRITE_SIZE) and dynamic instruction counts by kind.
TAG=${1:-r03zb}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SUM=gpurun_out/${TAG}_merit2_pmc.txt
: > $SUM
for v in 1 0; do
  export ALTRO_HIP_MERIT2_DPP=$v
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
    i=$((i + 1))
    OUT=gpurun_out/prof_${TAG}_d${v}_$i; rm -rf $OUT; mkdir -p $OUT
    echo "# ---- ALTRO_HIP_MERIT2_DPP=$v: rocprofv3 --kernel-trace --pmc $ctrs -- python tools/c1_solve.py 3" >> $SUM
    timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT -o t -- python tools/c1_solve.py 3 > $OUT/log.txt 2>&1
    python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep "counter \|merit" | grep -v "^void.*%$" | cut -c1-170 >> $SUM
    find $OUT -name "*.db" -delete
  done
done
cat $SUM
