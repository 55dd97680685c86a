#!/bin/bash
# N bench runs under torch.distributed.run (1 rank): does every one end with a JSON line?  (the NCCL watchdog thread polls its events while
# the library captures its graphs: profiles/r5/26_*)   usage: bash scripts/torchrun_soak.sh <runs> [steps]
cd $GRAFT_REPO_ROOT
ok=0; bad=0
for i in $(seq 1 ${1:-6}); do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + i)) bench.py --gpus 1 --steps ${2:-3} --warmup 1 > /tmp/soak_$i.json 2> /tmp/soak_$i.err
  if grep -q '"metric"' /tmp/soak_$i.json; then ok=$((ok + 1)); else bad=$((bad + 1)); echo "run $i failed: $(grep -h -m2 -E 'LadiffHipError|capturing|Error' /tmp/soak_$i.err | cut -c1-200)"; fi
done
echo "torchrun soak: $ok ok, $bad failed"
