#!/bin/bash
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
for v in _hdr10 _hdr9; do
  echo "== variant '$v'"
  STRL_BENCH_NOCHECK=1 STRL_LIB=$PWD/strling_amd/lib/libstrling_amd$v.so timeout 600 python tools/inflate_bench.py 524288 32768 2>&1 | grep "GB/s\|Error\|error"
done > gpurun_out/r4/inflate_exp3.txt 2>&1
cat gpurun_out/r4/inflate_exp3.txt
