#!/bin/bash
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bgzf_device.py tests/test_fast_inflate.py tests/test_front_device.py -m gpu -x -q 2>&1 | tail -3
for v in "" _w7 _w8; do
  echo "variant [$v]"
  STRL_LIB=$PWD/strling_amd/lib/libstrling_amd$v.so timeout 600 python tools/inflate_bench.py 524288 32768 2>&1 | tail -3 | head -2
done | tee gpurun_out/r4/inflate_vmatch_waves.txt
