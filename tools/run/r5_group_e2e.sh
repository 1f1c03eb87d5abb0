#!/bin/bash
# GPU box: `strling extract` on a 6.7e7-read level-6 BAM with the wave form and the grouped form of the device inflate: wall, loop, .bin identical.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R && export TMPDIR=/tmp
echo "== front-end tests, grouped form"
STRL_INFLATE_FORM=group timeout 900 python -m pytest tests/test_bgzf_device.py tests/test_front_device.py tests/test_regions_device.py -m gpu -x -q 2>&1 | tail -2
N=${N:-33554432}
python tools/e2e_bench.py $N --dir /tmp --check-slabs 0 --repeats 1 --keep --out /tmp/e2e_raw.json > /tmp/e2e_raw.log 2>&1
CLI=$R/strling_amd/lib/strling
cd /tmp
for rep in 1 2 3; do
  for f in wave group; do
    echo "-- $f (run $rep)"
    STRL_INFLATE_FORM=$f $CLI extract -v -g /tmp/e2e_${N}_6.str /tmp/e2e_${N}_6.bam /tmp/e2e_$f.bin 2>&1 | grep -i "seconds: total\|inflate" | cut -c1-400
  done
done
cmp /tmp/e2e_wave.bin /tmp/e2e_group.bin && echo ".bin identical (wave form = grouped form)"
