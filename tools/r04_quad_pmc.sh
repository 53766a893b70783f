#!/bin/bash
# tools/r04_quad_pmc.sh <tag> -- what bounds the four-lanes-per-problem sweeps at 8192 problems (C3 / C2 per-GPU share): cycle and
# instruction counters of quad_backward_kernel / quad2_backward_kernel / quad_forward_kernel under rocprofv3 --kernel-trace --pmc
TAG=${1:-r04zf}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_${TAG}_quad; rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/${TAG}_quad_pmc.txt; : > $SUM
for cfg in c3 c2; do
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    echo "# ---- $cfg: rocprofv3 --kernel-trace --pmc $ctrs -- python bench.py --config $cfg --batch 8192 --steps 10 --warmup 2 --no-cpu-baseline --sweeps-only --repeat-seconds 0" >> $SUM
    timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/${cfg}_$i -o t -- python bench.py --config $cfg --batch 8192 --steps 10 --warmup 2 --no-cpu-baseline --sweeps-only --repeat-seconds 0 > $OUT/${cfg}_$i.log 2>&1
    python tools/rocpd_summary.py $(find $OUT/${cfg}_$i -name "*.db") | grep "counter\|quad" | cut -c1-190 >> $SUM
  done
done
find $OUT -name "*.db" -delete
cat $SUM
