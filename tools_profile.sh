#!/bin/bash
# tools_profile.sh <tag> -- rocprofv3 kernel-trace stats + PMC passes for the bench command (GPU box).
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/trace_stdout.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/pmc_sq -o bench -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o bench -- $CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
find $OUT -name "*.csv" | head -50
ls -la $OUT/*
