#!/bin/bash
# the per-read state sized by the index's record counts: CLI / front end / shares tests, the 1.3e8-read file end to end with the
# slab check (one context, --gpus 4), the -v lines with and without the count
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6t; mkdir -p $O
CLI=$R/strling_amd/lib/strling
timeout 900 python -m pytest tests/test_cli.py tests/test_front_device.py tests/test_multi_device.py tests/test_call.py tests/test_verify_kit.py tests/test_pair_total.py -q -m gpu > $O/gpu_tests.txt 2>&1; grep -E 'passed|failed' $O/gpu_tests.txt | tail -2
timeout 900 python tools/e2e_bench.py 67108864 --dir /dev/shm --check-slabs 16 --repeats 2 --keep --out $O/e2e_2p27.json > $O/e2e_2p27.log 2>&1; tail -3 $O/e2e_2p27.log | cut -c1-600
timeout 600 python tools/e2e_bench.py 67108864 --dir /dev/shm --check-slabs 8 --gpus 4 --keep --out $O/e2e_2p27_g4.json > $O/e2e_2p27_g4.log 2>&1; tail -2 $O/e2e_2p27_g4.log | cut -c1-600
B=$(ls /dev/shm/e2e_67108864_*.bam | head -1); S=${B%.bam}.str
{
for how in A=1 STRL_NO_INDEX_COUNT=1 A=1 STRL_NO_INDEX_COUNT=1; do sleep 3; echo "== extract, $how"; ( time env $how STRL_ALLOC_TIMING=1 timeout 300 $CLI extract -v -g $S $B /dev/shm/x_$how.bin ) 2>&1 | grep -E 'records by the index|seconds before|real|hipMalloc (2[0-9]|1[0-9]|[5-9])\.' | cut -c1-420; done
cmp "/dev/shm/x_A=1.bin" "/dev/shm/x_STRL_NO_INDEX_COUNT=1.bin" && echo ".bin identical (sized by the index / by the file's bytes)"
} > $O/index_count_2p27.log 2>&1
cat $O/index_count_2p27.log
