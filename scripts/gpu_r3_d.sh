#!/bin/bash
TAG=${1:-r3g}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log; grep -E "oracle\| =|text tokens" $O/pytest.log | tail -30
timeout 600 python scripts/batch_scaling.py > $O/batch_scaling.log 2>&1; echo "batch_scaling rc=$?"; tail -9 $O/batch_scaling.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/c1trace -- python3 $GRAFT_REPO_ROOT/scripts/profile_pass.py bf16x3 6 c1 > $GRAFT_REPO_ROOT/$O/c1trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
f = glob.glob("$O/c1trace/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("c1 decode: total kernel time per pass %.1f us over 7 passes, launches per pass %.1f" % (tot / 7e3, sum(int(r["Calls"]) for r in rows) / 7))
for r in rows[:14]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"]) / 1e3:9.2f} us total {float(r["TotalDurationNs"]) / 1e6:8.3f} ms')
PY
