#!/bin/bash
# A/B of the C4 forward sweep on one box: the one-problem-per-wave kernel (fp64 arithmetic on fp32 records,
# ALTRO_HIP_F32_PURE_FWD_V1) and the four-problems-per-wave fp32 kernel's (ring depth, waves per SIMD) variants
# (ALTRO_HIP_F32X4_FWD=DW).   bash tools/c4_fwd_ab.sh [tag]
TAG=${1:-r03e}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -q -s -k "fp32 or c4 or 4block or fixture" 2>&1 | grep -E "C4 N=512|passed|failed|Error|assert" | tail -12) > gpurun_out/${TAG}_tests.log
for V in 14 24 22 32 42 23 33; do ALTRO_HIP_F32X4_FWD=$V timeout 200 python bench.py --config c4 --no-cpu-baseline --repeat-seconds 0.5 > gpurun_out/${TAG}_c4fwd_x4_$V.json 2>/dev/null; done
ALTRO_HIP_F32_PURE_FWD_V1=1 timeout 200 python bench.py --config c4 --no-cpu-baseline --repeat-seconds 0.5 > gpurun_out/${TAG}_c4fwd_v1.json 2>/dev/null
cat gpurun_out/${TAG}_tests.log
python - <<PY | tee gpurun_out/${TAG}_c4_fwd_variants.txt
import json,glob
print("# tools/c4_fwd_ab.sh: C4 (N=512, batch=16384, fp32) forward kernel variants, bench.py --config c4, one box; (avg, min) ms per launch")
for f in sorted(glob.glob('gpurun_out/${TAG}_c4fwd_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[0]); k=d['config']['kernels']
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {n:(round(v['avg_ms'],3),round(v['min_ms'],3)) for n,v in k.items()}, 'fwd frac %.3f'%d['roofline_forward']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY

# VMEM instructions per launch, before / after (one PMC pass each, --kernel-trace only)
for mode in v1 x4; do
  if [ $mode = v1 ]; then export ALTRO_HIP_F32_PURE_FWD_V1=1; else unset ALTRO_HIP_F32_PURE_FWD_V1; fi
  OUT=gpurun_out/prof_${TAG}_vmem_$mode; rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES -d $OUT -o t -- python bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline --repeat-seconds 0 > $OUT/log.txt 2>&1
  echo "# forward sweep, $mode: rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES (per launch)" >> gpurun_out/${TAG}_c4_fwd_variants.txt
  python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) | grep -E "forward" | cut -c1-200 >> gpurun_out/${TAG}_c4_fwd_variants.txt
  find $OUT -name "*.db" -delete
done
unset ALTRO_HIP_F32_PURE_FWD_V1
tail -30 gpurun_out/${TAG}_c4_fwd_variants.txt
