#!/bin/bash
# The one GPU-box runner.  gpurun -- 'bash scripts/gpu.sh <tag> <task> [<task> ...]'; output under gpurun_out/<tag>/.
# A task is name[:arg[:arg...]] (no spaces):
#   tests[:pytest -k expr]   pytest -m gpu                      smoke                __graft_entry__.smoke()
#   bench[:cfg[:steps]]      bench.py --config cfg              torchrun             the headline under torch.distributed.run, 1 rank
#   profile[:passes[:sha]]   kernel-trace + 3 PMC runs of scripts/profile_pass.py -> summary.md / summary.json
#   trace:<script>[:args..]  rocprofv3 --kernel-trace --stats of `python3 scripts/<script> args` -> top kernels
#   pmc:<kernel>:<ctrs,>:<script>[:args..]   PMC counters of one kernel
#   stamps[:B[:steps[:level]]]   diagnostic twin: per-block timeline / stage busy totals of the pipeline loop
#   py:<script>[:args..]     python scripts/<script> args > <script>.log
#   env:NAME=VALUE           export for the following tasks
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
top_kernels() {   # <stats dir> <n>
python3 - "$1" "${2:-24}" <<PY
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)
if not fs: sys.exit("no kernel_stats.csv under " + sys.argv[1])
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:int(sys.argv[2])]:
    print(f'{r["Name"][:84]:84s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"]) / 1e3:9.2f} us total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms')
PY
}
for task in "$@"; do
  IFS=: read -r name a1 a2 a3 rest <<< "$task"
  echo "=== $task"
  case $name in
    env) export "$a1";;
    tests)
      if [ -n "$a1" ]; then timeout 3000 python -m pytest tests -m gpu -x -q -k "$a1" > $O/pytest.log 2>&1; else timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; fi
      echo "pytest rc=$?"; tail -5 $O/pytest.log;;
    smoke) timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2;;
    bench)
      c=${a1:-headline}; timeout 900 python bench.py --config $c --steps ${a2:-20} --warmup 5 > $O/bench_${c}_n1.json 2> $O/bench_${c}_n1.err
      echo "bench $c rc=$?"; cut -c1-420 $O/bench_${c}_n1.json; tail -2 $O/bench_${c}_n1.err;;
    torchrun)
      timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_headline_torchrun_n1.json 2> $O/bench_headline_torchrun_n1.err
      echo "torchrun bench rc=$?"; cut -c1-200 $O/bench_headline_torchrun_n1.json;;
    gpus2)
      timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_gpus2_on_one_gpu.out 2> $O/bench_gpus2_on_one_gpu.err
      echo "plain --gpus 2 on a 1-GPU box rc=$? (expected non-zero: not enough devices)"; grep -h "not enough devices" $O/bench_gpus2_on_one_gpu.err | head -2;;
    profile)
      bash scripts/gpu_profile.sh $TAG/prof ${a2:-unknown} ${a1:-4} > $O/profile.log 2>&1
      echo "profile rc=$?"; grep -E "systolic|dec_mlp|dec_qkv|dec_out|Whole pass" $O/prof/summary.md;;
    trace)
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$a1 -- python3 $GRAFT_REPO_ROOT/scripts/$a1 $a2 $a3 ${rest//:/ } > $O/trace_$a1.log 2>&1)
      top_kernels $O/trace_$a1 ${TOPN:-24} | tee $O/trace_$a1.top.txt
      find $O/trace_$a1 -name "*.csv" -size +8M -delete;;
    pmc) bash scripts/gpu_pmc_kernel.sh $TAG/pmc_$a1 "$a1" "${a2//,/ }" -- scripts/$a3 ${rest//:/ } 2>&1 | tail -12;;
    stamps)
      lvl=${a3:-2}
      LADIFF_STAMPS_LEVEL=$lvl timeout 900 python scripts/stamps_pipeline.py ${a1:-128} ${a2:-50} pipeline16 2>&1 | grep -v amdgpu.ids > $O/stamps_b${a1:-128}_l$lvl.log
      tail -${TAILN:-60} $O/stamps_b${a1:-128}_l$lvl.log | cut -c1-260;;
    py)
      log=$O/${a1%.py}${LOGSUFFIX}.log
      timeout ${PY_TIMEOUT:-1200} python scripts/$a1 $a2 $a3 ${rest//:/ } 2>&1 | grep -v amdgpu.ids > $log; echo "rc=${PIPESTATUS[0]}"; tail -${TAILN:-40} $log | cut -c1-300;;
    *) echo "unknown task $name";;
  esac
done
