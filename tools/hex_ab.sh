#!/bin/bash
# sixteen lanes per problem (ALTRO_HIP_LANE_HEX=1) against the round-4 sweeps (=0) of the (4, 2) and (2, 1) shapes over the batch size
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05a}_hex_ab.txt
echo "# bench.py --config c3|c2 --batch B, ALTRO_HIP_LANE_HEX=1|0: (backward, forward) kernel ms per launch" > $O
for cfg in c3 c2; do
for B in ${2:-2048 4096 8192 16384 32768 65536}; do
  for Q in 1 0; do
    ALTRO_HIP_LANE_HEX=$Q python bench.py --config $cfg --batch $B --repeat-seconds 0 --steps 20 --no-other-configs --no-live-traffic --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config']['kernels']
print('$cfg batch %6d hex=$Q  ms/step %.4f  ' % ($B, d['ms_per_step']), {n:round(v['avg_ms'],4) for n,v in k.items()})
" >> $O
  done
done
done
cat $O
