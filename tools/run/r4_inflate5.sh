#!/bin/bash
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
for v in "" _r9w6; do
  echo "== variant '$v'"
  STRL_LIB=$PWD/strling_amd/lib/libstrling_amd$v.so timeout 600 python tools/ubench/inflate_symbols.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r4/inflate_symbols.txt 2>&1
cat gpurun_out/r4/inflate_symbols.txt
