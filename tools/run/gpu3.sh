cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export STRL_FRONT_TIMING=1
timeout 900 python tools/e2e_bench.py 8388608 2>&1 | tail -5
