#!/bin/bash
# A/B of the pure-fp32 C4 backward kernels on one box: the one-problem-per-wave kernel (ALTRO_HIP_F32_PURE_V1) and the
# four-problems-per-wave kernel's (ring depth, waves per SIMD) variants (ALTRO_HIP_F32X4=DW).   bash tools/c4_ab.sh [tag]
TAG=${1:-r02c}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -q -x -k "fp32 or c4 or 4block or fixture" 2>&1 | tail -5) > gpurun_out/${TAG}_tests.log
for V in 13 22 32 31; do ALTRO_HIP_F32X4=$V timeout 200 python bench.py --config c4 --no-cpu-baseline --repeat-seconds 0.5 > gpurun_out/${TAG}_c4pure_x4_$V.json 2>/dev/null; done
ALTRO_HIP_F32_PURE_V1=1 timeout 200 python bench.py --config c4 --no-cpu-baseline --repeat-seconds 0.5 > gpurun_out/${TAG}_c4pure_v1.json 2>/dev/null
cat gpurun_out/${TAG}_tests.log
python - <<PY | tee gpurun_out/${TAG}_c4_quad_variants.txt
import json,glob
print("# tools/c4_ab.sh: C4 (N=512, batch=16384, fp32) backward kernel variants, bench.py --config c4, one box; (avg, min) ms per launch")
for f in sorted(glob.glob('gpurun_out/${TAG}_c4pure_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[0]); k=d['config']['kernels']
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {n:(round(v['avg_ms'],3),round(v['min_ms'],3)) for n,v in k.items()}, 'frac %.3f'%d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
