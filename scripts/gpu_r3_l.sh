#!/bin/bash
TAG=${1:-r3l}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "final_layer or fused_mlp or decode or in_proj_inside" > $O/pytest_k.log 2>&1
echo "pytest kernels rc=$?"; tail -3 $O/pytest_k.log
timeout 600 python scripts/attn_speed.py > $O/attn_speed.log 2>&1; tail -4 $O/attn_speed.log
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_abi.py -m gpu -x -q > $O/pytest_path.log 2>&1
echo "pytest path rc=$?"; tail -3 $O/pytest_path.log
