#!/bin/bash
# tools/profile_quad13.sh <tag> [batch] -- rocprofv3 kernel-trace stats of tools/quad13_nmpc_time.py (13-state quadrotor NMPC on plan MFMA32)
TAG=${1:-r06o}; B=${2:-4096}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/prof_${TAG}_quad13; rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/${TAG}_quad13_nmpc_$B.txt
echo "# python tools/quad13_nmpc_time.py $B 6  (rocprofv3 --kernel-trace --stats, MI355X)" > $SUM
timeout 300 python tools/quad13_nmpc_time.py $B 6 | tail -3 >> $SUM
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python tools/quad13_nmpc_time.py $B 6 > $OUT/trace.log 2>&1
python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep -v "rocclr\|_pack_kernel\|unpack_kernel" | cut -c1-170 | head -16 >> $SUM
find $OUT -name "*.db" -delete
cat $SUM
