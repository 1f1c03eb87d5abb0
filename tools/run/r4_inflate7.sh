#!/bin/bash
# throughput of the inflate kernel against waves per CU (unused dynamic LDS lowers the occupancy; 9-bit tables: 4.4 KB static)
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
for pad in 36000 15500 8900 5600 3600 2200 1200 0; do
  echo "== lds pad $pad (waves per CU by LDS: $((163840 / (4608 + pad))))"
  STRL_INFLATE_LDS_PAD=$pad timeout 600 python tools/inflate_bench.py 524288 32768 2>&1 | grep "GB/s"
done > gpurun_out/r4/inflate_occupancy.txt 2>&1
for v in _w7 _w8; do echo "== variant $v"; STRL_LIB=$PWD/strling_amd/lib/libstrling_amd$v.so timeout 600 python tools/inflate_bench.py 524288 32768 2>&1 | grep "GB/s"; done >> gpurun_out/r4/inflate_occupancy.txt 2>&1
cat gpurun_out/r4/inflate_occupancy.txt
