#!/bin/bash
# round end: whole GPU suite, smoke, one bench line per BASELINE config (+ e2e, + under torchrun), the profiling runs -> summary.json
TAG=${1:-r4end}; SHA=${2:-unknown}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
grep -h "max\|err" $O/pytest.log | head -0
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_headline_n1.json 2> $O/bench_headline_n1.err
echo "bench rc=$?"; cut -c1-200 $O/bench_headline_n1.json; tail -2 $O/bench_headline_n1.err
for c in c1 c2 c3 c5 e2e; do
  timeout 900 python bench.py --config $c > $O/bench_${c}_n1.json 2> $O/bench_${c}_n1.err; echo "bench $c rc=$?"; cut -c1-160 $O/bench_${c}_n1.json
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_headline_torchrun_n1.json 2> $O/bench_headline_torchrun_n1.err
echo "torchrun bench rc=$?"; cut -c1-160 $O/bench_headline_torchrun_n1.json
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_gpus2_on_one_gpu.out 2> $O/bench_gpus2_on_one_gpu.err; echo "plain --gpus 2 on a 1-GPU box rc=$? (expected non-zero: not enough devices)"; grep -h "not enough devices" $O/bench_gpus2_on_one_gpu.err | head -2
bash scripts/gpu_profile.sh $TAG/prof $SHA 4 > $O/profile.log 2>&1
echo "profile rc=$?"; grep -E "systolic|dec_mlp|Whole pass" $O/prof/summary.md
timeout 300 python scripts/decode_speed.py > $O/decode_speed.log 2>&1; tail -4 $O/decode_speed.log
timeout 600 python scripts/handoff_knobs.py tags= 2>&1 | grep -v amdgpu.ids > $O/handoff_rates.log; cat $O/handoff_rates.log
timeout 600 python scripts/pass_ab.py 2>&1 | grep -v amdgpu.ids > $O/pass_ab.log; cat $O/pass_ab.log
