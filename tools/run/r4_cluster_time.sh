#!/bin/bash
# where `merge` / `call` spend their clustering seconds at whole-genome size: a .bin of ~8e6 treads tiled from a smaller extract
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s5
mkdir -p $O
cd $R
python tools/e2e_bench.py 33554432 --dir /tmp --check-slabs 0 --repeats 1 --keep --out $O/e2e_d.json > $O/e2e_d.log 2>&1
python - <<'P'
import numpy as np, sys
sys.path.insert(0, '.')
from strling_amd import api
b = api.bin_read('/tmp/e2e_33554432_6.bin')
t = np.tile(b['treads'], 8)
n0 = len(b['treads'])
t['qname_id'] = np.tile(np.arange(n0), 8)
api.bin_write('/tmp/big.bin', 0.8, 40, b['frag'], b['header'], t, b['qname_off'], b['qnames'])
print(len(t), 'treads')
P
CLI=$R/strling_amd/lib/strling
for k in 1 2; do
  sleep 2
  { time STRL_CLUSTER_TIMING=1 $CLI merge -v -o /tmp/mrg /tmp/big.bin ; } 2>&1 | grep "strl_cluster\|cluster_collect\|seconds:\|real" >> $O/cluster_time.txt
done
