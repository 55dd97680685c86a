#!/bin/bash
TAG=${1:-r3h}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -x -q -k "fused_mlp or vae_decode or ragged or encode" > $O/pytest_dec.log 2>&1
echo "pytest dec rc=$?"; tail -5 $O/pytest_dec.log
timeout 300 python scripts/decode_speed.py > $O/decode_speed.log 2>&1; echo "decode_speed rc=$?"; tail -12 $O/decode_speed.log
