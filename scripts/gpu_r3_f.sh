#!/bin/bash
# full GPU suite + headline bench + kernel stats
TAG=${1:-r3m}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_headline_n1.json 2> $O/bench_headline_n1.err
echo "bench headline rc=$?"; cut -c1-200 $O/bench_headline_n1.json; tail -2 $O/bench_headline_n1.err
bash scripts/gpu_trace.sh $TAG/trace 4 2>&1 | tail -24
