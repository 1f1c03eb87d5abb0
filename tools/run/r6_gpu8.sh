#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
CLI=$R/strling_amd/lib/strling
python - > $O/make_2p27.log 2>&1 <<'PY'
import sys; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import e2e_bench
inp = e2e_bench.make_input(67108864, d='/dev/shm')
PY
B=/dev/shm/e2e_67108864_6
$CLI extract -g $B.str $B.bam $B.bin 2> /dev/null
for how in device host device host; do
  sleep 3; echo "== call, STRL_CALL_FRAG=$how"
  ( time STRL_CALL_FRAG=$how STRL_FRAG_TIMING=1 STRL_CLUSTER_TIMING=1 STRL_BIN_TIMING=1 $CLI call -v -o /dev/shm/c_$how $B.bam $B.bin ) 2>&1 | grep -E 'seconds:|real|strl_bin_read|cluster_collect\]|strl_cluster\]|on the host|fragment lengths\]' | cut -c1-330
done > $O/call_frag_ab.log 2>&1
cat $O/call_frag_ab.log
for f in bounds genotype unplaced; do cmp /dev/shm/c_device-$f.txt /dev/shm/c_host-$f.txt && echo "$f identical"; done
