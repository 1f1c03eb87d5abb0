#!/bin/bash
# decode-only thread sweep of the host BAM reader (strling _decode) on a synthetic BAM
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python - <<'PY'
import sys; sys.path.insert(0, "tools")
import e2e_bench
print(e2e_bench.make_input(int(sys.argv[1]) if len(sys.argv) > 1 else 2**23))
PY
for t in 8 16 32 64 128; do
  STRL_THREADS=$t STRL_DECODE_TIMING=1 strling_amd/lib/strling _decode /tmp/e2e_8388608.bam 2>&1 | grep -v "^$" | tr '\n' ' '; echo
done
for t in 32 64; do
  echo "taskset node0, $t threads:"; STRL_THREADS=$t STRL_DECODE_TIMING=1 taskset -c 0-63,128-191 strling_amd/lib/strling _decode /tmp/e2e_8388608.bam 2>&1 | tr '\n' ' '; echo
done
