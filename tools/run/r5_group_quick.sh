#!/bin/bash
# GPU box: parity of the grouped inflate (G from $GS, default 8) against zlib, then its throughput on the two block sets.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
for g in ${GS:-8}; do
  echo "== group form G=$g"
  STRL_INFLATE_FORM=group STRL_INFLATE_G=$g timeout 600 python -m pytest tests/test_bgzf_device.py -x -q -m gpu 2>&1 | tail -2
  STRL_INFLATE_FORM=group STRL_INFLATE_G=$g timeout 900 python tools/inflate_bench.py $((1<<19)) ${MINB:-32768} 2>&1 | tail -3
done
