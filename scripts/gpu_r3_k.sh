#!/bin/bash
# attention with in_proj inside: tests, decode timing, kernel stats of a decode
TAG=${1:-r3k}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O



cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/scripts/decode_prof.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/decode_kernel_stats.csv; head -14 $O/decode_kernel_stats.csv | cut -c1-60,100-200
