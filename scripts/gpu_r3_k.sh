#!/bin/bash
# attention with in_proj inside: tests, decode timing, kernel stats of a decode
TAG=${1:-r3k}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "final_layer or fused_mlp or decode or in_proj_inside" > $O/pytest_k.log 2>&1
echo "pytest kernels rc=$?"; tail -3 $O/pytest_k.log
timeout 600 python scripts/decode_speed.py > $O/decode_speed.log 2>&1; grep -E "fused_mlp=(1|17):" $O/decode_speed.log
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o dec -- python3 $GRAFT_REPO_ROOT/scripts/decode_prof.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/decode_kernel_stats.csv; head -14 $O/decode_kernel_stats.csv | cut -c1-60,100-200
