#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== sharded tool alone"; timeout 300 python tools/sharded_c_abi.py 1 100000 10 2>&1 | tail -2
echo "== bench under torchrun, 1 rank, exchange forced"; HS_BENCH_FORCE_EXCHANGE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/bench_dist.json 2> gpurun_out/bench_dist.err; echo "rc=$?"; tail -3 gpurun_out/bench_dist.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_dist.json').read().strip().splitlines()[-1])
print(d['value'], d['n_gpus'])
print(json.dumps(d.get('sharded_committee'))[:700])
print(json.dumps(d.get('sharded_c_abi'))[:1200])
PY
