#!/bin/bash
# the pair pass's buffers allocated with the per-read state at bring-up instead of behind the loop: -v lines of the full-size extract
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6l; mkdir -p $O
CLI=$R/strling_amd/lib/strling
timeout 900 python -m pytest tests/test_cli.py tests/test_front_device.py tests/test_gpu_parity.py tests/test_pair_total.py tests/test_long_reads.py -q -m gpu -x > $O/tests_10.txt 2>&1; tail -2 $O/tests_10.txt | cut -c1-200
python - > $O/make_full.log 2>&1 <<'PY'
import sys; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import e2e_bench
inp = e2e_bench.make_input(268435456, d='/tmp')
PY
B=$(ls /tmp/e2e_268435456_6.bam /dev/shm/e2e_268435456_6.bam 2>/dev/null | head -1); S=${B%.bam}.str; D=$(dirname $B)
{
for rep in 1 2 3 4; do
  sleep 6; echo "== extract, run $rep"
  ( time STRL_FRONT_TIMING=1 timeout 300 $CLI extract -v -g $S $B $D/x_$((rep % 2)).bin ) 2>&1 | grep -E 'seconds: total|seconds before|process:|real|treads_named' | cut -c1-700
done
cmp $D/x_0.bin $D/x_1.bin && echo ".bin identical"
sleep 6; echo "== extract, allocation timing"
( time STRL_ALLOC_TIMING=1 timeout 300 $CLI extract -v -g $S $B $D/x_0.bin ) 2>&1 | grep -E 'alloc|seconds: total|real' | cut -c1-300 | tail -40
for rep in 1 2 3; do sleep 5; echo "== call, run $rep"; ( time STRL_BIN_TIMING=1 timeout 300 $CLI call -v -o $D/c $B $D/x_1.bin ) 2>&1 | grep -E 'seconds:|real|strl_bin_read' | cut -c1-500; done
for rep in 1 2; do sleep 5; echo "== merge, run $rep"; ( time STRL_BIN_TIMING=1 timeout 300 $CLI merge -v -o $D/m $D/x_1.bin ) 2>&1 | grep -E 'seconds:|real|strl_bin_read' | cut -c1-400; done
} > $O/pair_prealloc_full_size.log 2>&1
cat $O/pair_prealloc_full_size.log
rm -f $D/x_*.bin $D/c-*.txt $D/m-*.txt
