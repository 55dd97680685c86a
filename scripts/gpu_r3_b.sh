#!/bin/bash
# round-3 GPU batch B: the new kernel tests first, then the whole suite, decoder timing with / without the fused feed-forward block
TAG=${1:-r3b}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_mlp" > $O/pytest_mlp.log 2>&1
echo "pytest mlp rc=$?"; tail -15 $O/pytest_mlp.log
timeout 300 python scripts/decode_speed.py > $O/decode_speed.log 2>&1; echo "decode_speed rc=$?"; cat $O/decode_speed.log | tail -12
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_headline_n1.json 2> $O/bench_headline_n1.err
echo "bench headline rc=$?"; cut -c1-300 $O/bench_headline_n1.json; tail -3 $O/bench_headline_n1.err
