#!/bin/bash
# round 5: the two fuzzers on the final build, whole output kept (master seeds, per-N-cases progress lines with the last case's seed and parameters, totals)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
timeout 900 python tests/fuzz/fuzz_parity.py ${1:-240} 2026 > gpurun_out/r5/fuzz_parity.log 2>&1; tail -2 gpurun_out/r5/fuzz_parity.log
timeout 900 python tests/fuzz/fuzz_parity.py ${1:-240} 505 > gpurun_out/r5/fuzz_parity_seed505.log 2>&1; tail -1 gpurun_out/r5/fuzz_parity_seed505.log
timeout 1200 python tests/fuzz/fuzz_call.py ${2:-420} 77 > gpurun_out/r5/fuzz_call.log 2>&1; tail -2 gpurun_out/r5/fuzz_call.log
