#!/bin/bash
TAG=${1:-r4c}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python scripts/handoff_ab.py 3,2,u 64,50,u 128,50,u 128,50,m 256,50,u > $O/handoff_ab.log 2>&1; echo "ab rc=$?"; grep "bit for bit\|tags :\|flags:" $O/handoff_ab.log
