#!/bin/bash
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bgzf_device.py tests/test_front_device.py tests/test_inflate_emu.py tests/test_cli.py -m gpu -x -q > gpurun_out/r4/t_inflate4.txt 2>&1; tail -3 gpurun_out/r4/t_inflate4.txt
timeout 600 python tools/inflate_bench.py 524288 32768 2>&1 | grep "GB/s\|rror" | tee gpurun_out/r4/inflate_unroll.txt
timeout 600 python tools/ubench/inflate_symbols.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/inflate_unroll.txt
timeout 900 python tests/fuzz/fuzz_call.py 300 2>&1 | tail -3 | tee gpurun_out/r4/fuzz_call.log
