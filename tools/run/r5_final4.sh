#!/bin/bash
# GPU box, last run of round 5: gpu suite, inflate fuzzer, inflate bench + counters, extract / call / merge on the 6.7e7-read file -- of the final build.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final4
mkdir -p $O
cd $R && export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt | tail -2
timeout 100 python tests/fuzz/fuzz_inflate.py 75 51 > $O/fuzz_inflate.log 2>&1; tail -1 $O/fuzz_inflate.log
timeout 100 python tests/fuzz/fuzz_call.py 60 91 > $O/fuzz_call.log 2>&1; tail -1 $O/fuzz_call.log
timeout 300 python tools/inflate_bench.py 524288 32768 > $O/inflate.log 2>&1; tail -1 $O/inflate.log > $O/inflate.json; grep "GB/s" $O/inflate.log
STRL_BENCH_NOCHECK=1 STRL_INFLATE_FORM=wave timeout 300 bash tools/prof_inflate.sh > $O/inflate_pmc.log 2>&1; grep "^L6 SQ_INSTS_SALU\|^L6 SQ_INSTS_VALU\|^L6 SQ_WAVE_CYCLES" $O/inflate_pmc.log
timeout 600 python tools/e2e_bench.py 33554432 --dir /tmp --check-slabs 2 --repeats 3 --out $O/e2e_2p26_reads.json > $O/e2e.log 2>&1; tail -3 $O/e2e.log | cut -c1-300
