#!/bin/bash
# round-3 first GPU pass: whole GPU suite, default bench line, two-rank bench forms through the gloo hook
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log
tail -5 $O/gputests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
ALTRO_BENCH_BACKEND=gloo python bench.py --gpus 2 --no-cpu-baseline --repeat-seconds 1 > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2_gloo.err; echo "bench2 rc=$?"
ALTRO_BENCH_BACKEND=gloo python bench.py --gpus 2 --config c3 --repeat-seconds 1 > $O/bench_c3_gpus2_gloo.json 2> $O/bench_c3_gpus2_gloo.err; echo "bench c3 rc=$?"
python bench.py --gpus 2 > $O/bench_gpus2_refused.json 2> $O/bench_gpus2_refused.err; echo "bench refused rc=$? (expected nonzero on a 1-GPU box)"
tail -2 $O/bench_gpus2_refused.err
cut -c1-600 $O/bench_default.json
