#!/bin/bash
# usage (on the GPU box): bash tools/lane_latency.sh  -> gpurun_out/lane_latency.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lane_latency.txt; : > $out
for cfg in "64 1 1" "8192 1 1" "8192 0 1" "8192 1 0" "65536 1 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace -d gpurun_out/ll_$tag -o t -- python tools/lane_latency.py $cfg > /dev/null 2>&1
  echo "### batch deriv al = $cfg" >> $out
  python tools/rocpd_summary.py $(find gpurun_out/ll_$tag -name "*.db" | head -1) | grep -v "rocclr\|pack\|copy" | cut -c1-150 >> $out
done
