#!/bin/bash
# pipeline tests + loop timing
TAG=${1:-r3n}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > $O/pytest_pipe.log 2>&1
echo "pytest pipeline rc=$?"; tail -4 $O/pytest_pipe.log
timeout 600 python scripts/try_pipeline.py uniform 128,50 256,50 > $O/try_uniform.log 2>&1; grep -E "pipeline16|pipeline32|max" $O/try_uniform.log | tail -8
timeout 600 python scripts/try_pipeline.py 128,50 > $O/try_mixed.log 2>&1; grep -E "pipeline16|max" $O/try_mixed.log | tail -4
timeout 600 python scripts/try_pipeline.py fp32 uniform 128,50 > $O/try_fp32.log 2>&1; grep -E "pipeline16|max" $O/try_fp32.log | tail -4
