#!/bin/bash
TAG=${1:-r4a}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python scripts/handoff_ab.py > $O/handoff_ab.log 2>&1; echo "ab rc=$?"; cat $O/handoff_ab.log | tail -40
timeout 300 python scripts/handoff_ab.py fp32 3,2,u 128,50,u 128,50,m > $O/handoff_ab_fp32.log 2>&1; echo "ab fp32 rc=$?"; tail -12 $O/handoff_ab_fp32.log
