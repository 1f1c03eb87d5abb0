cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^$" | grep -i "passed\|failed\|error" | tail -3
