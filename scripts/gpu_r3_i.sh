#!/bin/bash
# fused decoder feed-forward kernel work: its tests + timing
TAG=${1:-r3i}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "final_layer or fused_mlp or decode or in_proj_inside" > $O/pytest_k.log 2>&1
echo "pytest kernels rc=$?"; tail -4 $O/pytest_k.log
timeout 300 python scripts/mlp_speed.py > $O/mlp_speed.log 2>&1; tail -3 $O/mlp_speed.log
timeout 600 python scripts/decode_speed.py > $O/decode_speed.log 2>&1; grep "fused_mlp=1" $O/decode_speed.log
grep "fused_mlp=17" $O/decode_speed.log
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_abi.py -m gpu -x -q > $O/pytest_path.log 2>&1
echo "pytest path rc=$?"; tail -4 $O/pytest_path.log
