#!/bin/bash
# round 4: the symbol loop on the vector unit -- correctness on the device, then the inflate kernel alone (level 1 / level 6 blocks)
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bgzf_device.py tests/test_front_device.py tests/test_inflate_emu.py tests/test_cli.py -m gpu -x -q > gpurun_out/r4/t_inflate.txt 2>&1; tail -5 gpurun_out/r4/t_inflate.txt
timeout 600 python tools/inflate_bench.py 524288 32768 > gpurun_out/r4/inflate_valu.txt 2>&1; tail -4 gpurun_out/r4/inflate_valu.txt
