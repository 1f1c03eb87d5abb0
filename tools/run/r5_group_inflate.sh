#!/bin/bash
# GPU box: the grouped inflate (inflate_group.h) against zlib and against the wave form: parity tests for G = 8 / 4 / 16, then throughput.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
for g in 8 4 16; do
  echo "== parity, group form G=$g"
  STRL_INFLATE_FORM=group STRL_INFLATE_G=$g timeout 600 python -m pytest tests/test_bgzf_device.py -x -q -m gpu 2>&1 | tail -3
done
echo "== throughput"
for f in wave group; do
  for g in 8 ${EXTRA_G}; do
    [ $f = wave ] && [ $g != 8 ] && continue
    echo "-- form=$f G=$g"
    STRL_INFLATE_FORM=$f STRL_INFLATE_G=$g timeout 900 python tools/inflate_bench.py $((1<<19)) $((1<<15)) 2>&1 | tail -4
  done
done
