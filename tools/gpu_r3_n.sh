#!/bin/bash
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "4 16" "0 16" "2 16" "8 16" "4 12" "4 8" "1 16" "4 16"; do
  set -- $cfg
  timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu --no-extras --partitions $1 --streams $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('partitions $1 streams $2: %.2f M  %s %.4f ms prepare %s' % (j['value']/1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare']))"
done
