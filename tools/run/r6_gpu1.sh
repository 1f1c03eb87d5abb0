#!/bin/bash
# round 6, first GPU call: the new long-read tests, then the whole GPU suite, then a quick bench line (no e2e) for the step
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_long_reads.py -x -q -m gpu > gpurun_out/r6/long_reads.txt 2>&1
tail -15 gpurun_out/r6/long_reads.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6/gpu_tests_1.txt 2>&1
tail -5 gpurun_out/r6/gpu_tests_1.txt
