#!/bin/bash
TAG=${1:-r4b}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python scripts/handoff_ab.py 3,2,u 7,5,m 64,50,u 128,50,u 128,50,m 256,50,u > $O/handoff_ab.log 2>&1; echo "ab rc=$?"; grep "bit for bit\|tags :" $O/handoff_ab.log
timeout 300 python scripts/handoff_ab.py fp32 3,2,u 128,50,u 128,50,m > $O/handoff_ab_fp32.log 2>&1; echo "ab fp32 rc=$?"; grep "bit for bit" $O/handoff_ab_fp32.log
timeout 600 python scripts/stamps_pipeline.py 128 50 pipeline16 > $O/stamps128.log 2>&1; tail -28 $O/stamps128.log
