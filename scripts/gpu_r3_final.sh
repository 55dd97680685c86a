#!/bin/bash
# round end: whole GPU suite, smoke, headline bench (as the driver runs it), the profiling runs -> profiles/r3/summary.json
TAG=${1:-r3end}; SHA=${2:-unknown}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_headline_n1.json 2> $O/bench_headline_n1.err
echo "bench rc=$?"; cut -c1-220 $O/bench_headline_n1.json; tail -2 $O/bench_headline_n1.err
bash scripts/gpu_profile.sh $TAG/prof $SHA 4 > $O/profile.log 2>&1
echo "profile rc=$?"; grep -E "systolic|dec_mlp|Whole pass" $O/prof/summary.md
