#!/bin/bash
# tools/r05_hex_pmc.sh <tag> [batch] -- cycle / instruction / LDS counters of hex_backward_kernel (and whatever else the sweep runs)
# at one batch size: rocprofv3 --kernel-trace --pmc, one counter group per pass
TAG=${1:-r05e}; BATCH=${2:-8192}; CFGS=${3:-c3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_${TAG}_hex; rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/${TAG}_hex_pmc.txt; : > $SUM
for cfg in $CFGS; do
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    echo "# ---- $cfg batch $BATCH: --pmc $ctrs" >> $SUM
    timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/${cfg}_$i -o t -- python bench.py --config $cfg --batch $BATCH --steps 10 --warmup 2 --no-cpu-baseline --sweeps-only --repeat-seconds 0 --no-other-configs --no-live-traffic > $OUT/${cfg}_$i.log 2>&1
    python tools/rocpd_summary.py $(find $OUT/${cfg}_$i -name "*.db") | grep "counter\|backward" | cut -c1-190 >> $SUM
  done
done
find $OUT -name "*.db" -delete
cat $SUM
