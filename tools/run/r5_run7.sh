#!/bin/bash
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 2400 python -u -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r5/t7_all.txt 2>&1; tail -6 gpurun_out/r5/t7_all.txt
for k in 1 2 3; do timeout 600 python -u -m pytest tests/test_cli.py tests/test_front_device.py -m gpu -q --timeout 150 > gpurun_out/r5/t7_rep$k.txt 2>&1; tail -2 gpurun_out/r5/t7_rep$k.txt; done
