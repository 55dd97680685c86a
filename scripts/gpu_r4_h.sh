#!/bin/bash
# kernel stats of the end-to-end pass (CLIP text tower included)
O=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --config e2e --steps 4 --warmup 1 --cpu-sample 2 > $O/stats.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -24 $f | cut -c1-170
