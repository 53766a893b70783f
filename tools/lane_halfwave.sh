#!/bin/bash
# Does a wave with only its lower 32 lanes active run fp64 VALU work faster on MI355X (SIMD-32)?  LANE kernel durations for
# one wave at 64 / 32 / 16 active lanes.   bash tools/lane_halfwave.sh -> gpurun_out/lane_halfwave.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lane_halfwave.txt; : > $out
for b in 64 32 16 8; do
  timeout 200 rocprofv3 --kernel-trace -d gpurun_out/lh_$b -o t -- python tools/lane_latency.py $b 1 1 > /dev/null 2>&1
  echo "### batch (active lanes of the one wave) = $b" >> $out
  python tools/rocpd_summary.py $(find gpurun_out/lh_$b -name "*.db" | head -1) | grep "lane_backward\|ilqr_merit\|lane_forward" | cut -c1-150 >> $out
  find gpurun_out/lh_$b -name "*.db" -delete
done
cat $out
