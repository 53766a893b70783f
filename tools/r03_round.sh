#!/bin/bash
# round-3 evidence pass: whole GPU suite, default bench (+ under rocprofv3 with the PMC passes), the other configs' lines,
# C1 solve profile, shape table.   bash tools/r03_round.sh <tag>
TAG=${1:-r03h}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log; tail -3 $O/gputests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for c in c2 c3 c4; do python bench.py --config $c --repeat-seconds 1 > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?"; done
python bench.py --config c3 --batch 8192 --repeat-seconds 1 > $O/bench_c3_8192.json 2> $O/bench_c3_8192.err
bash tools/profile_default.sh $TAG > /dev/null 2>&1
bash tools/r03_c1_solve_profile.sh $TAG > /dev/null 2>&1
python tools/shape_cliff.py > $O/shape_cliff.txt 2>&1
ls -la gpurun_out/${TAG}_* $O
