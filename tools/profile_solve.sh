#!/bin/bash
# tools/profile_solve.sh <tag> [configs...] -- rocprofv3 kernel-trace stats of the batched SOLVE of the LANE configurations
# on the launch-sequenced loop (ALTRO_HIP_FUSED=0) and on the policy's choice; summaries in gpurun_out/<tag>_<cfg>_<mode>.txt
TAG=${1:-r02}; shift
CFGS=${@:-"c2 c3"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in $CFGS; do
  for mode in sequenced policy; do
    CMD="python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --repeat-seconds 0"
    OUT=gpurun_out/prof_${TAG}_${cfg}_$mode
    rm -rf $OUT; mkdir -p $OUT
    SUM=gpurun_out/${TAG}_${cfg}_solve_$mode.txt
    if [ $mode = sequenced ]; then export ALTRO_HIP_FUSED=0; else unset ALTRO_HIP_FUSED; fi
    echo "# ALTRO_HIP_FUSED=${ALTRO_HIP_FUSED:-unset} $CMD  (rocprofv3 --kernel-trace --stats, MI355X)" > $SUM
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
    grep '^{' $OUT/trace.log >> $SUM
    python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep -v "rocclr\|expand_copy\|gather_copy\|_pack_kernel\|unpack_kernel" | cut -c1-200 >> $SUM
    find $OUT -name "*.db" -delete
  done
done
unset ALTRO_HIP_FUSED
ls -la gpurun_out/${TAG}_*solve*.txt
