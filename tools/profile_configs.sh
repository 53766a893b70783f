#!/bin/bash
# tools/profile_configs.sh <tag> [configs...] -- rocprofv3 kernel-trace stats + PMC passes (each counter set in its own
# pass, kernel-trace only) for the non-headline bench configurations; summaries land in gpurun_out/<tag>_<cfg>.txt.
#   bash tools/profile_configs.sh r02a c4 c4pure c2 c3
TAG=${1:-r02}; shift
CFGS=${@:-"c4 c4mixed c2 c3"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in $CFGS; do
  case $cfg in
    c1) ARGS="--config c1" ;;
    c4) ARGS="--config c4" ;;
    c4mixed) ARGS="--config c4 --c4-mixed" ;;
    *) ARGS="--config $cfg" ;;
  esac
  CMD="python bench.py $ARGS --steps 5 --warmup 1 --no-cpu-baseline --repeat-seconds 0"
  OUT=gpurun_out/prof_${TAG}_$cfg
  rm -rf $OUT; mkdir -p $OUT
  SUM=gpurun_out/${TAG}_$cfg.txt
  echo "# $CMD  (rocprofv3, MI355X; one pass per counter set, --kernel-trace only)" > $SUM
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
  grep '^{' $OUT/trace.log >> $SUM
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
              "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
              "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
              "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/pmc$i -o t -- $CMD > $OUT/pmc$i.log 2>&1   # (bounded: a pass that hangs must not eat the box)
  done
  python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep -v "rocclr\|expand_copy\|gather_copy\|_pack_kernel\|unpack_kernel" | cut -c1-220 >> $SUM
  find $OUT -name "*.db" -delete   # summaries are what travels back (gpurun_out is capped at 64 MiB)
done
ls -la gpurun_out/${TAG}_*.txt
