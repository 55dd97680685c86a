#!/bin/bash
# GPU box: kernel trace of the benchmark pass, per-kernel average durations.  bash scripts/gpu_trace.sh <out_dir under gpurun_out> [passes]
OUT=$1; PASSES=${2:-3}
ROOTD=$GRAFT_REPO_ROOT
mkdir -p $ROOTD/gpurun_out/$OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/$OUT/stats -- python3 $ROOTD/scripts/profile_pass.py bf16x3 $PASSES > $ROOTD/gpurun_out/$OUT/stats.log 2>&1
cd $ROOTD
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/$OUT/stats/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"]) / 1e3:9.2f} us total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms')
PY
