#!/bin/bash
# GPU box: kernel-trace + PMC runs of one precision mode of the benchmark pass, summarised into profiles/<round>/ files.
#   bash scripts/gpu_profile.sh <out_dir under gpurun_out> <git sha> [passes]
OUT=$1; SHA=$2; PASSES=${3:-2}
ROOTD=$GRAFT_REPO_ROOT
mkdir -p $ROOTD/gpurun_out/$OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/$OUT/stats -- python3 $ROOTD/scripts/profile_pass.py f16x3 $PASSES > $ROOTD/gpurun_out/$OUT/stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $ROOTD/gpurun_out/$OUT/SQ_VALU_MFMA_BUSY_CYCLES -- python3 $ROOTD/scripts/profile_pass.py f16x3 $PASSES > $ROOTD/gpurun_out/$OUT/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOTD/gpurun_out/$OUT/FETCH_SIZE -- python3 $ROOTD/scripts/profile_pass.py f16x3 $PASSES > $ROOTD/gpurun_out/$OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOTD/gpurun_out/$OUT/WRITE_SIZE -- python3 $ROOTD/scripts/profile_pass.py f16x3 $PASSES > $ROOTD/gpurun_out/$OUT/write.log 2>&1
cd $ROOTD
python3 scripts/pmc_summary.py gpurun_out/$OUT $PASSES gpurun_out/$OUT/summary.md gpurun_out/$OUT/summary.json $SHA
tail -3 gpurun_out/$OUT/*.log
head -30 gpurun_out/$OUT/summary.md
# keep only what is small: the raw counter CSVs of a 77-launch-per-step run are tens of MB
find gpurun_out/$OUT -name "*_counter_collection.csv" -size +8M -delete
find gpurun_out/$OUT -name "*_kernel_trace.csv" -size +8M -delete
