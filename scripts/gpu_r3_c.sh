#!/bin/bash
TAG=${1:-r3c}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_mlp" > $O/pytest_mlp.log 2>&1
echo "pytest mlp rc=$?"; tail -5 $O/pytest_mlp.log
timeout 300 python scripts/mlp_speed.py > $O/mlp_speed.log 2>&1; echo "mlp_speed rc=$?"; tail -8 $O/mlp_speed.log
