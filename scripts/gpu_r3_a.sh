#!/bin/bash
# round-3 GPU batch A: the whole GPU test suite, one bench line per BASELINE config, the torchrun N=1 line, the four profiling runs.
#   bash scripts/gpu_r3_a.sh <tag> <git sha>
TAG=${1:-r3a}; SHA=${2:-unknown}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log; grep -E "max \|frames|oracle\| =" $O/pytest.log | tail -20
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_headline_n1.json 2> $O/bench_headline_n1.err
echo "bench headline rc=$?"; cut -c1-400 $O/bench_headline_n1.json
for c in c1 c2 c3 c5; do
  timeout 900 python bench.py --gpus 1 --config $c > $O/bench_${c}_n1.json 2> $O/bench_${c}_n1.err
  echo "bench $c rc=$?"; cut -c1-330 $O/bench_${c}_n1.json; tail -2 $O/bench_${c}_n1.err
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_headline_torchrun_n1.json 2> $O/bench_headline_torchrun_n1.err
echo "torchrun rc=$?"; cut -c1-200 $O/bench_headline_torchrun_n1.json
bash scripts/gpu_profile.sh $TAG/prof $SHA 4 > $O/profile.log 2>&1
echo "profile rc=$?"; tail -32 $O/profile.log
