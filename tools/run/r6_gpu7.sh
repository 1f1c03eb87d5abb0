#!/bin/bash
# round 6, seventh GPU call: `strling call` with the fragment-length sample through the device, the .bin reader and clustering host
# side trimmed: the call tests, then extract / call / merge on the 1.3e8-read file (call with STRL_CALL_FRAG=host beside it)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
CLI=$R/strling_amd/lib/strling
timeout 900 python -m pytest tests/test_call.py tests/test_regions_device.py tests/test_cli.py -q -m gpu -x > $O/call_tests_7.txt 2>&1; tail -3 $O/call_tests_7.txt | cut -c1-300
python tools/e2e_bench.py 67108864 --dir /dev/shm --keep --check-slabs 4 --repeats 2 --out $O/e2e_2p27_call.json > $O/e2e_2p27_call.log 2>&1
python - <<'PY'
import json
e = json.load(open('gpurun_out/r6/e2e_2p27_call.json'))
print('extract', e.get('extract_s'), 'call', e.get('call_s'), 'merge', e.get('merge_s'), 'check', (e.get('check') or {}).get('ok'))
print(' call:', e.get('call_phases', '')[:500]); print(' merge:', e.get('merge_phases'))
PY
B=/dev/shm/e2e_67108864_6
for how in device host device host; do
  sleep 3; echo "== call, STRL_CALL_FRAG=$how"
  ( time STRL_CALL_FRAG=$how STRL_CLUSTER_TIMING=1 STRL_BIN_TIMING=1 $CLI call -v -o /dev/shm/c_$how $B.bam $B.bin ) 2>&1 | grep -E 'seconds:|real|strl_bin_read|cluster_collect\]|strl_cluster\]|on the host' | cut -c1-420
done > $O/call_frag_ab.log 2>&1
cat $O/call_frag_ab.log
for f in bounds genotype unplaced; do cmp /dev/shm/c_device-$f.txt /dev/shm/c_host-$f.txt && echo "$f identical"; done
