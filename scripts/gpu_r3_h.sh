#!/bin/bash
# final_layer on bf16x3 tiles: tests + decode timing + headline
TAG=${1:-r3h}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "final_layer or fused_mlp or decode" > $O/pytest_k.log 2>&1
echo "pytest kernels rc=$?"; tail -4 $O/pytest_k.log
timeout 600 python scripts/decode_speed.py > $O/decode_speed.log 2>&1; cat $O/decode_speed.log | tail -12
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_abi.py -m gpu -x -q > $O/pytest_path.log 2>&1
echo "pytest path rc=$?"; tail -4 $O/pytest_path.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-400
