#!/bin/bash
# usage (on the GPU box): bash tools/lane_latency2.sh  -> gpurun_out/lane_latency2.txt   (batch sweep, both LANE models)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lane_latency2.txt; : > $out
for cfg in "64 1 0 100 pendulum" "1024 1 0 100 pendulum" "8192 1 0 100 pendulum" "64 1 1 50 bicycle" "1024 1 1 50 bicycle" "8192 1 1 50 bicycle"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace -d gpurun_out/ll_$tag -o t -- python tools/lane_latency.py $cfg > /dev/null 2>&1
  echo "### batch deriv al N model = $cfg" >> $out
  python tools/rocpd_summary.py $(find gpurun_out/ll_$tag -name "*.db" | head -1) | grep -v "rocclr\|pack\|copy" | cut -c1-150 >> $out
  rm -rf gpurun_out/ll_$tag
done
