#!/bin/bash
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bgzf_device.py tests/test_fast_inflate.py tests/test_front_device.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python tools/inflate_bench.py 524288 32768 2>&1 | tail -4 | tee gpurun_out/r4/inflate_sentinel.txt
