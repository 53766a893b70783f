#!/bin/bash
# tools/tile32_pmc.sh <tag> <n,m> ... -- what bounds plan MFMA32's sweep kernels: rocprofv3 --kernel-trace --stats, then cycle /
# instruction / LDS counters (separate --pmc passes) of tile32_backward_kernel / tile32_forward_kernel at 4096 problems x 128 knot points
TAG=${1:-r06a}; shift
SHAPES=${@:-13,4}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_${TAG}_tile32; rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/${TAG}_tile32_pmc.txt; : > $SUM
for shp in $SHAPES; do
  echo "# ==== shape $shp: rocprofv3 --kernel-trace --stats -- python tools/tile32_check.py --time-only $shp" >> $SUM
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/s_$shp -o t -- python tools/tile32_check.py --time-only $shp > $OUT/s_$shp.log 2>&1
  cat $OUT/s_$shp.log | grep "N=" >> $SUM
  python tools/rocpd_summary.py $(find $OUT/s_$shp -name "*.db") | grep "tile32\|kernel" | cut -c1-200 >> $SUM
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM" "FETCH_SIZE WRITE_SIZE"; do
    i=$((i+1))
    echo "# ---- $shp: --pmc $ctrs" >> $SUM
    timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/p${i}_$shp -o t -- python tools/tile32_check.py --time-only $shp > $OUT/p${i}_$shp.log 2>&1
    python tools/rocpd_summary.py $(find $OUT/p${i}_$shp -name "*.db") | grep "counter\|tile32" | cut -c1-220 >> $SUM
  done
done
find $OUT -name "*.db" -delete
cat $SUM
