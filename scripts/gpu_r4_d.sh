#!/bin/bash
TAG=${1:-r4k}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python scripts/handoff_knobs.py tags= 2>&1 | grep -v amdgpu.ids | tee $O/rates.log
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > $O/pytest_pipeline.log 2>&1; echo "pytest pipeline rc=$?"; tail -5 $O/pytest_pipeline.log
