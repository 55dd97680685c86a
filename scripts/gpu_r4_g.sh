#!/bin/bash
# kernel stats of the decode alone (default switches)
O=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/scripts/decode_prof.py > $O/stats.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-160
