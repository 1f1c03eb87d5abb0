#!/bin/bash
# the hybrid (scalar + vector) symbol loop: correctness on the device, then the kernel alone, 10-bit and 9-bit first level
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bgzf_device.py tests/test_front_device.py tests/test_inflate_emu.py tests/test_cli.py -m gpu -x -q > gpurun_out/r4/t_inflate2.txt 2>&1; tail -5 gpurun_out/r4/t_inflate2.txt
for v in "" _r9w6; do
  echo "== variant '$v'"
  STRL_LIB=$PWD/strling_amd/lib/libstrling_amd$v.so timeout 600 python tools/inflate_bench.py 524288 32768 2>&1 | grep "GB/s\|Error\|error"
  STRL_LIB=$PWD/strling_amd/lib/libstrling_amd$v.so timeout 600 python tools/ubench/inflate_symbols.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r4/inflate_hybrid.txt 2>&1
cat gpurun_out/r4/inflate_hybrid.txt
