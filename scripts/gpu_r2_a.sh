#!/bin/bash
# round-2 GPU batch A: torchrun N=1 (RCCL initialised, final all-gather executed), plain N=1, new parity tests, MFMA calibration
set -x
mkdir -p gpurun_out/r2a
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/r2a/bench_torchrun_n1.json 2> gpurun_out/r2a/bench_torchrun_n1.err
echo "torchrun rc=$?"
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/r2a/bench_plain_n1.json 2> gpurun_out/r2a/bench_plain_n1.err
echo "plain rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r2a/pytest.log
echo "pytest rc=$?"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/calib -- $GRAFT_REPO_ROOT/scripts/ubench_mfma_calib.bin > $GRAFT_REPO_ROOT/gpurun_out/r2a/calib.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2a/calib.err
echo "calib rc=$?"
cd $GRAFT_REPO_ROOT
tail -5 gpurun_out/r2a/pytest.log
cat gpurun_out/r2a/bench_torchrun_n1.json | cut -c1-300
cat gpurun_out/r2a/bench_plain_n1.json | cut -c1-300
cat gpurun_out/r2a/calib.json
