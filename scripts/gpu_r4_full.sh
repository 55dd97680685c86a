#!/bin/bash
TAG=${1:-r4full}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -8 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_headline.json 2> $O/bench_headline.err; echo "bench rc=$?"; cut -c1-700 $O/bench_headline.json
