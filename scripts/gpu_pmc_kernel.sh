#!/bin/bash
# PMC counters of one kernel: bash scripts/gpu_pmc_kernel.sh <tag> <kernel-name-substring> <counters (quoted)> -- <python script + args>
TAG=$1; KEY=$2; CTRS=$3; shift 4
ROOTD=$GRAFT_REPO_ROOT
mkdir -p $ROOTD/gpurun_out/$TAG
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $ROOTD/gpurun_out/$TAG/pmc -- python3 $ROOTD/$@ > $ROOTD/gpurun_out/$TAG/pmc.log 2>&1
cd $ROOTD
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/$TAG/pmc/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); n = collections.defaultdict(int); dur = 0.0; nd = 0
for r in csv.DictReader(open(f)):
    if "$KEY" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    if r["Counter_Name"] == list(acc)[0]:
        dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; nd += 1
print("$TAG $KEY: launches", nd, "avg us", dur / max(nd, 1))
for k in acc: print(f"  {k:32s} {acc[k] / n[k]:16.0f} per launch")
PY
find gpurun_out/$TAG -name "*.csv" -size +4M -delete
