#!/bin/bash
# SQ / TCP counters of the C4 pure-fp32 backward kernel variant $1 (ALTRO_HIP_F32X4=DW, or "v1"):  bash tools/c4_pmc.sh 22 tag
V=${1:-22}; TAG=${2:-r02e}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$V" = "v1" ]; then export ALTRO_HIP_F32_PURE_V1=1; else export ALTRO_HIP_F32X4=$V; fi
CMD="python bench.py --config c4 --c4-pure --no-cpu-baseline --steps 3 --warmup 1 --repeat-seconds 0"
OUT=gpurun_out/prof_${TAG}_$V; rm -rf $OUT; mkdir -p $OUT
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/pmc$i -o t -- $CMD > $OUT/pmc$i.log 2>&1
done
python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep -i "backward\|^kernel" | cut -c1-200 > gpurun_out/${TAG}_c4pure_pmc_$V.txt
find $OUT -name "*.db" -delete
cat gpurun_out/${TAG}_c4pure_pmc_$V.txt | awk '{print $(NF-7), $(NF-6), $(NF-5), $(NF-4)}' | column -t | head -80
