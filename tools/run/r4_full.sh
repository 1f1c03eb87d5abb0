#!/bin/bash
# round 4: the headline configuration at its size -- 2^29 distinct reads (5.4e8, a 26x genome of 2.7 Gbp), zlib level 6, binned qualities, aux tags
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 2400 python tools/e2e_bench.py $((1<<28)) --check-slabs 64 --repeats 3 --out gpurun_out/r4/e2e_full.json > gpurun_out/r4/e2e_full.log 2>&1
tail -c 6000 gpurun_out/r4/e2e_full.log
