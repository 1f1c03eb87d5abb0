#!/bin/bash
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 1200 python -u -m pytest tests/test_cli.py tests/test_cram.py tests/test_front_device.py tests/test_abi.py tests/test_multi_device.py -m gpu -v --timeout 150 > gpurun_out/r5/t6.txt 2>&1; grep -n 'PASSED\|FAILED\|Timeout\|ERROR' gpurun_out/r5/t6.txt | tail -70; tail -5 gpurun_out/r5/t6.txt
