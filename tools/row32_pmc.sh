#!/bin/bash
# tools/row32_pmc.sh "<n m ...>" -- kernel times and SQ / TA / TCP counters of a whole solve at a shape (tools/solve_shapes.py)
SHAPE=${1:-"13 4"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=/tmp/row32_pmc; rm -rf $OUT; mkdir -p $OUT
CMD="python tools/solve_shapes.py $SHAPE"
rocprofv3 --kernel-trace --stats -d $OUT/t -o t -- $CMD > $OUT/t.log 2>&1
python tools/rocpd_summary.py $(find $OUT/t -name "*.db") | grep "merit\|kernel  " | cut -c1-150
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$OUT/p_$(echo $set | cut -c1-12 | tr ' ' '_'); mkdir -p $d
  rocprofv3 --kernel-trace --pmc $set -d $d -o t -- $CMD > $d/log 2>&1
  python tools/rocpd_summary.py $(find $d -name "*.db") | grep "merit" | grep -v "^void\|^altro" | cut -c1-160
  python tools/rocpd_summary.py $(find $d -name "*.db") | awk '/counter/{f=1} f' | grep "merit" | cut -c1-170
done
