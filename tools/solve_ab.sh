#!/bin/bash
# fused vs launch-sequenced solve on the C2 / C3 batches (bench.py's full_solve leg), same box.  bash tools/solve_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in c2 c3; do
  for mode in fused sequenced; do
    if [ $mode = sequenced ]; then export ALTRO_HIP_NO_FUSED=1; else unset ALTRO_HIP_NO_FUSED; fi
    timeout 200 python bench.py --config $c --no-cpu-baseline --repeat-seconds 0.2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c $mode', d['config']['full_solve'])"
  done
done
